"""GPU: 2-D END-TO-END parity at the reference network size (VERDICT r4 missing #3).

(i)  FREE-RUNNING, multi-step: the reference's Kolmogorov net (experiments/kolmogorov/train.py:15-22 -- window 5, (96, 192, 384),
     (3, 3, 3), 22.9 M parameters) at 64 x 64 on trajectories of L = 6 (two windows each), Gaussian guidance through
     ``x[..., ::4, ::4]`` with std 0.1, one Langevin correction per step -- 32 diffusion steps = 64 guided evaluations deep, every
     step replayed from the captured hipGraph, nothing teacher-forced -- against the oracle's sampling loop (sda/score.py:225-263,
     :375-396) run in fp32 and in fp64 from the same initial draw and the same corrector noise.  Bound (SURVEY 8c tier 3): the
     reference arithmetic's own fp32-vs-fp64 deviation on this chain, floor 1e-4 -- and that deviation must itself stay small,
     otherwise the chain is too ill-conditioned to test anything and the test FAILS rather than passing vacuously.
(ii) STATISTICAL, the reference's own acceptance check for this experiment -- ``(A(x) - y_star).std()  # should be ~ 0.1``
     (experiments/kolmogorov/figures.ipynb#cell11) -- in the form tests/test_gpu_lorenz_eval.py has for Lorenz: with
     ``bench.SyntheticScore(net, scale=0)`` (the network still runs, forward and VJP, in every evaluation) the prior is N(0, I) and
     eps is exact, so the posterior given ``y = A x + N(0, std^2)`` is Gaussian in closed form: N(y / (1 + std^2),
     std^2 / (1 + std^2)) on the observed pixels, N(0, 1) elsewhere.  Full K64 net at 64 x 64, 16 trajectories x 16 frames,
     128 steps x C = 1, free-running on the device RNG; checked against the closed form where the reference ALGORITHM attains it and
     against two independent runs of the oracle's loop for what the algorithm itself biases.
"""
import math
import os

import pytest
import torch

from oracle import sda_oracle as O
from tests.util import oracle_eps_from_module, rel_err

pytestmark = pytest.mark.gpu
# The oracle runs on the host: one guided K64 evaluation at 64 x 64 costs it ~2.5 s in fp32 and ~5 s in fp64, so the routine suite keeps
# the chains short (8 free-running steps = 16 guided evaluations deep; 64 trajectories x 128 steps for the statistics: ~3 min in all).
# SDA_LONG_TESTS=1 runs the long forms (32 steps, B = 1 and 2; 256 trajectories x 128 steps: ~25 min of host time) -- green on the
# round-5 build, profiles/r05_gputest_237_long_variants.log.  Round 6: the 32-step chains run in EVERY `-m gpu` pass against oracle
# trajectories committed as data (test_k64_32_step_chain_vs_committed_oracle_trajectories; tests/golden/make_golden_k64_chain.py).
LONG = os.environ.get('SDA_LONG_TESTS', '0') == '1'
K64 = dict(window=5, embedding=64, hidden_channels=(96, 192, 384), hidden_blocks=(3, 3, 3))


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def k64(dev):
    from sda_amd.experiments.kolmogorov import make_score
    torch.manual_seed(70)
    net = make_score(size=64, **K64)
    assert sum(p.numel() for p in net.parameters()) == 22_874_922          # the reference K64 net
    return net, oracle_eps_from_module(net, 'mc2d')


def _sub4(x):
    return x[..., ::4, ::4]


@pytest.mark.parametrize('batch', [1, 2] if LONG else [1])
def test_k64_free_running_guided_hipgraph_vs_oracle(dev, k64, batch):
    import bench
    from sda_amd import observe as Ob
    from sda_amd.parallel import KeyedNoise
    from sda_amd.score import GaussianScore, VPSDE
    net, eps_net = k64
    net.to(dev)
    steps, corr, tau, std, gamma = (32 if LONG else 8), 1, 0.5, 0.1, 1e-2
    event = (6, 2, 64, 64)
    torch.manual_seed(71 + batch)
    x1 = torch.randn((batch,) + event)
    y = torch.randn(_sub4(x1[0]).shape)                         # one observation shared by the batch (figures.ipynb#cell10)
    ns = KeyedNoise((0, batch), event, 77, corr, dev)           # graph-safe corrector noise; the same draws go to the oracle
    zs = torch.stack([ns(i, j) for i in range(steps) for j in range(corr)]).cpu()

    score = bench.SyntheticScore(net)                           # a raw random-init net overflows under guidance
    inner = VPSDE(score, shape=())
    object.__setattr__(score, '_sched', inner)
    gs = GaussianScore(y, A=Ob.Subsample.space(4), std=std, sde=inner, gamma=gamma)
    sde = VPSDE(gs, shape=event).to(dev)
    sde.initial_noise, sde.noise_source = x1, ns
    sampler = sde.sampler((batch,), steps=steps, corrections=corr, tau=tau).capture()
    assert sampler._graph is not None
    for _ in range(steps):
        sampler.step()
    got = sampler.result().cpu()
    assert torch.isfinite(got).all()

    sched = O.Schedule()

    def oracle(dtype):
        def eps(xx, tt):
            mu, sg = sched.mu(tt), sched.sigma(tt)
            return xx * (sg / (mu * mu + sg * sg)) + 0.1 * eps_net(xx, tt, None if dtype == torch.float32 else dtype)
        sc = lambda xx, tt: O.gaussian_score(eps, sched, y.to(dtype), _sub4, std, gamma, xx, tt)
        zz = zs.to(dtype)
        return O.sample(sc, sched, x1.to(dtype), 4, steps, corr, tau, noise=lambda i, j: zz[i * corr + j])

    ref32, ref64 = oracle(torch.float32), oracle(torch.float64)
    assert torch.isfinite(ref64).all()
    own = rel_err(ref32.double(), ref64)
    err = rel_err(got.double(), ref64)
    err32 = rel_err(got.double(), ref32.double())
    print(f'K64 @ 64^2, B = {batch}, L = 6, {steps} guided PC steps (C = 1) through the hipGraph: HIP vs fp64 oracle {err:.2e}, '
          f'HIP vs fp32 oracle {err32:.2e}, fp32 oracle vs fp64 oracle {own:.2e}')
    assert own < 2e-3, (f'the oracle disagrees with itself across precisions by {own:.2e} on this chain: too ill-conditioned for a '
                        f'free-running parity test to mean anything -- shorten it instead of widening the bound')
    # north_star's own bar, not elastic: the HIP path against the fp32 reference arithmetic from the same draws, rtol 1e-4 (measured
    # 2e-5 after 64 guided evaluations: the two fp32 chains stay together although each is 2e-4 from the fp64 one)
    assert err32 <= 1e-4, f'{steps}-step free-running guided sample: HIP path vs the fp32 oracle {err32:.2e} > 1e-4'
    assert err <= max(1e-4, 3 * own), (f'{steps}-step free-running guided sample: HIP path vs fp64 oracle {err:.2e}; the fp32 oracle itself '
                                       f'is {own:.2e} from the fp64 oracle (bound: max(1e-4, 3x that))')


def _digest(t):
    d = t.detach().double().reshape(-1)
    w = torch.arange(1, d.numel() + 1, dtype=torch.float64) % 8191 + 1
    return torch.tensor([d.sum().item(), d.abs().sum().item(), (d * w).sum().item()], dtype=torch.float64)


def _chain_inputs(batch, steps, corr=1, event=(6, 2, 64, 64)):
    """Exactly tests/golden/make_golden_k64_chain.py::chain_inputs: everything from seeds, the corrector noise from a HOST generator."""
    torch.manual_seed(71 + batch)
    x1 = torch.randn((batch,) + event)
    y = torch.randn(x1[0][..., ::4, ::4].shape)
    g = torch.Generator().manual_seed(7700 + batch)
    zs = torch.randn((steps * corr, batch) + event, generator=g)
    return x1, y, zs


@pytest.mark.parametrize('batch', [2, 1])
def test_k64_32_step_chain_vs_committed_oracle_trajectories(dev, k64, batch):
    """The LONG form of the test above in every routine run: 32 free-running guided PC steps (64 guided K64
    evaluations deep) replayed from the hipGraph, against the oracle's fp32 and fp64 final samples of the SAME chain computed in the
    build container and committed as data (tests/golden/k64_chain_b*.npz, made by make_golden_k64_chain.py from oracle/ alone: ~20 min
    of host time per trajectory that the GPU box no longer spends).  Net, x(1), y and the recorded corrector noise are regenerated
    from the seeds here; their float64 digests must equal the fixture's, else the fixture is stale and the test says so."""
    import bench
    import numpy as np
    from sda_amd import observe as Ob
    from sda_amd.parallel import TableNoise
    from sda_amd.score import GaussianScore, VPSDE
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', f'k64_chain_b{batch}.npz')
    if not os.path.exists(path):
        pytest.skip(f'{os.path.basename(path)} not generated (python3 -B tests/golden/make_golden_k64_chain.py --batches {batch})')
    fx = np.load(path)
    steps, corr, tau, std, gamma = int(fx['steps']), 1, 0.5, 0.1, 1e-2
    net, _ = k64
    x1, y, zs = _chain_inputs(batch, steps, corr)
    params = torch.cat([p.detach().reshape(-1).cpu() for p in net.parameters()])
    for name, t in (('params', params), ('x1', x1), ('y', y), ('zs', zs)):
        want, got_d = torch.from_numpy(fx[name + '_digest']), _digest(t)
        assert torch.allclose(got_d, want, rtol=1e-12, atol=0), \
            (f'stale fixture {os.path.basename(path)}: {name} regenerated from its seed differs from what the oracle ran on '
             f'(torch {torch.__version__} here, {fx["torch_version"]} there) -- regenerate the fixture')
    net.to(dev)
    event = (6, 2, 64, 64)
    score = bench.SyntheticScore(net)
    inner = VPSDE(score, shape=())
    object.__setattr__(score, '_sched', inner)
    gs = GaussianScore(y, A=Ob.Subsample.space(4), std=std, sde=inner, gamma=gamma)
    sde = VPSDE(gs, shape=event).to(dev)
    sde.initial_noise, sde.noise_source = x1, TableNoise(zs.to(dev), corr)
    sampler = sde.sampler((batch,), steps=steps, corrections=corr, tau=tau).capture()
    assert sampler._graph is not None
    mid = None
    for i in range(steps):
        sampler.step()
        if i == 7:
            mid = sampler.result().cpu().clone()
    got = sampler.result().cpu()
    assert torch.isfinite(got).all()
    ref32, ref64 = torch.from_numpy(fx['ref32']), torch.from_numpy(fx['ref64'])
    own = float(fx['own_fp32_vs_fp64'])
    err, err32 = rel_err(got.double(), ref64), rel_err(got.double(), ref32.double())
    err8 = rel_err(mid.double(), torch.from_numpy(fx['ref32_step8']).double())
    print(f'K64 @ 64^2, B = {batch}, L = 6, {steps} guided PC steps (C = 1) through the hipGraph vs the committed oracle trajectories: '
          f'HIP vs fp64 oracle {err:.2e}, vs fp32 oracle {err32:.2e} (after 8 steps {err8:.2e}), fp32 oracle vs fp64 oracle {own:.2e}')
    assert own < 2e-3, f'the oracle disagrees with itself across precisions by {own:.2e}: the chain is too ill-conditioned to test anything'
    assert err8 <= 1e-4 and err32 <= 1e-4, f'HIP path vs the fp32 oracle: {err8:.2e} after 8 steps, {err32:.2e} after {steps} (> 1e-4)'
    assert err <= max(1e-4, 3 * own), f'HIP path vs fp64 oracle {err:.2e}; the fp32 oracle itself is {own:.2e} away (bound max(1e-4, 3x))'


_ORACLE_RUNS = {}


def _oracle_samples(y, std, gamma, seed, shape, steps, corr, tau):
    key = (seed, tuple(shape), steps, corr, tau, std, gamma, float(y.double().sum()))
    if key not in _ORACLE_RUNS:                                           # (shared by the eager and the graph variant)
        _ORACLE_RUNS[key] = _oracle_samples_run(y, std, gamma, seed, shape, steps, corr, tau)
    return _ORACLE_RUNS[key]


def _oracle_samples_run(y, std, gamma, seed, shape, steps, corr, tau):
    sched = O.Schedule()
    eta = 1e-3
    eps = lambda x, t: x * (sched.sigma(t) / (1 + eta * eta))            # exact for N(0, I) data (bench.SyntheticScore, scale = 0)
    score = lambda x, t: O.gaussian_score(eps, sched, y, _sub4, std, gamma, x, t)
    torch.manual_seed(seed)
    return O.sample(score, sched, torch.randn(shape), 4, steps, corr, tau)


def _per_sample_log_spread(obs, y):
    return (obs - y).flatten(1).std(dim=1).log()


@pytest.mark.parametrize('graph', [False, True])
def test_kolmogorov_assimilation_statistical_end_to_end(dev, k64, graph):
    import bench
    from sda_amd import observe as Ob
    from sda_amd.score import GaussianScore, VPSDE
    import numpy as np
    net, _ = k64
    net.to(dev)
    # round 6: the oracle side of the 256-trajectory form is committed as data (tests/golden/k64_stats_b256.npz: the per-trajectory
    # log-spreads and the two scalars of the two oracle runs, made by make_golden_k64_chain.py --stats 256), so every routine run
    # has the long form's statistical power; without the fixture: 64 trajectories, oracle runs on this host (256 with SDA_LONG_TESTS)
    fpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'k64_stats_b256.npz')
    fx = np.load(fpath) if os.path.exists(fpath) else None
    long_form = LONG or fx is not None
    B, L, steps, corr, tau, std, gamma = (256 if long_form else 64), 6, 128, 1, 0.5, 0.1, 1e-2      # (128 steps: at 64 the reference ALGORITHM's own slope is 0.986, not 0.990)
    event = (L, 2, 64, 64)
    torch.manual_seed(80)
    y = torch.randn(_sub4(torch.empty(event)).shape) * math.sqrt(1 + std ** 2)        # y = A x + noise, x ~ N(0, I)
    if fx is not None:
        assert torch.allclose(_digest(y), torch.from_numpy(fx['y_digest']), rtol=1e-12, atol=0), 'stale fixture k64_stats_b256.npz (y)'
        ref_a = ref_b = None
    else:
        ref_a = _oracle_samples(y, std, gamma, 81, (B,) + event, steps, corr, tau)
        ref_b = _oracle_samples(y, std, gamma, 82, (B,) + event, steps, corr, tau)

    score = bench.SyntheticScore(net, scale=0.0)                          # the net still runs (forward + VJP) at every evaluation
    inner = VPSDE(score, shape=())
    object.__setattr__(score, '_sched', inner)
    gs = GaussianScore(y, A=Ob.Subsample.space(4), std=std, sde=inner, gamma=gamma)
    sde = VPSDE(gs, shape=event).to(dev)
    sde.use_graph = graph
    torch.manual_seed(83)
    x = sde.sample((B,), steps=steps, corrections=corr, tau=tau).cpu()
    assert torch.isfinite(x).all()

    zcrit = 4.5
    obs = _sub4(x)                                                        # (B, L, 2, 16, 16)
    n = obs.numel()
    pm = y / (1 + std ** 2)
    # (a) analytic posterior mean on the observed pixels (the corrector inflates their variance, not their mean): pooled z-score
    #     with the sample's own spread, and the regression slope of the samples on y (closed form 1 / (1 + std^2))
    res = obs - pm
    z = (res.mean() / (res.std() / math.sqrt(n))).abs().item()
    assert z < zcrit, f'observed pixels: pooled mean {z:.1f} standard errors from the analytic posterior mean'
    yy = y.expand_as(obs)
    slope = ((obs * yy).sum() / (yy * yy).sum()).item()
    se = (res.std() / (yy * yy).sum().sqrt()).item()
    assert abs(slope - 1 / (1 + std ** 2)) < zcrit * se + 1e-4, f'slope of A(x) on y {slope:.5f} vs {1 / (1 + std ** 2):.5f} (se {se:.1e})'
    # (b) the reference's own acceptance check, figures.ipynb#cell11: (A(x) - y).std().  Measured with the oracle's loop: on this
    #     N(0, I) prior the reference ALGORITHM leaves ~ 3 std with one tau = 0.5 correction per step (a predictor-only run collapses
    #     to 0.1 std), and the per-trajectory spread is log-normal-ish with sd(log) ~ 0.44, because each trajectory adapts its own
    #     Langevin step (score.py:259).  So: same order as std, and the per-trajectory log-spreads of the GPU run against those of
    #     an oracle run by a two-sample z-test -- with the second oracle run as a check that the test is calibrated
    mask = torch.ones(64, 64, dtype=torch.bool)
    mask[::4, ::4] = False
    if fx is not None:
        spread_a, un_var_a = float(fx['spread_a']), float(fx['unobserved_var_a'])
        la, lb = torch.from_numpy(fx['log_spread_a']), torch.from_numpy(fx['log_spread_b'])
    else:
        oa, ob = _sub4(ref_a), _sub4(ref_b)
        spread_a, un_var_a = (oa - y).std().item(), ref_a[..., mask].var().item()
        la, lb = _per_sample_log_spread(oa, y), _per_sample_log_spread(ob, y)
    spread = (obs - y).std().item()
    assert 0.5 * std < spread < 6 * std, f'(A(x) - y).std() = {spread:.4f} is not of the order of std = {std}'
    lg = _per_sample_log_spread(obs, y)
    two = lambda p, q: ((p.mean() - q.mean()) / (p.var() / B + q.var() / B).sqrt()).abs().item()
    z_s, z_cal = two(lg, la), two(lb, la)
    assert z_cal < zcrit, f'oracle run vs oracle run: {z_cal:.1f} sigma -- the yardstick itself is off'
    assert z_s < zcrit, (f'per-trajectory log (A(x) - y).std(): GPU {lg.mean():.3f} vs oracle {la.mean():.3f} ({z_s:.1f} sigma; oracle pair '
                         f'{z_cal:.1f} sigma)')
    assert abs(lg.std().item() / la.std().item() - 1) < (0.25 if long_form else 0.4)          # and the same dispersion (sd of the ratio ~ 1 / sqrt(B))
    # (c) unobserved pixels keep the prior N(0, 1): mean, and variance as the oracle's loop leaves it
    un = x[..., mask]
    assert abs(un.mean().item()) < zcrit / math.sqrt(un.numel())
    assert abs(un.var().item() / un_var_a - 1) < 0.01
    print(f'kolmogorov assimilation (graph={graph}): (A(x)-y).std() {spread:.4f} (oracle {spread_a:.4f}; std {std}), mean z {z:.2f}, '
          f'slope {slope:.5f} (closed form {1 / (1 + std ** 2):.5f}), log-spread {lg.mean():.3f} +- {lg.std():.3f} (oracle '
          f'{la.mean():.3f} +- {la.std():.3f}; z {z_s:.2f}, oracle pair {z_cal:.2f})')
