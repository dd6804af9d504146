"""GPU: the reference's canonical caller as a test -- experiments/lorenz/eval.py:72-84:
``sde.sample((1024,), steps=256, corrections=C, tau=0.25)`` on event (65, 3) with ``GaussianScore(y, A = x[..., ::step, :1], std,
gamma=3e-2)``, step / std = 8 / 0.05 ("lo") and 1 / 0.25 ("hi"), for the global (lorenz/utils.py:26-42) and the local
(:45-59) score networks.

(i) one guided evaluation at B = 1024, L = 65 (odd length, 16 x the whole-net kernel grid of BASELINE configs[1]) against the
    oracle on a 32-row slice (the guidance gradient never couples rows);
(ii) a STATISTICAL end-to-end check, the way the reference validates itself (eval.py:57-94: moments / W1 of sample sets against a
    second sample set; figures.ipynb#cell11: (A(x) - y).std() ~ std): with ``bench.SyntheticScore(net, scale=0)`` the data
    distribution is N(0, I) and eps is exact, so the posterior is closed-form Gaussian -- N(y / (1 + std^2), std^2 / (1 + std^2)) on
    the observed components, N(0, 1) elsewhere.  1024 samples x 256 steps x C = 1 on the GPU, free-running, against (a) the analytic
    posterior where the reference ALGORITHM itself attains it (the mean; the unobserved components) and (b) two independent
    1024-sample runs of the oracle's sampling loop (sda/score.py:225-263) for everything the algorithm biases (measured here with
    the oracle: the Langevin corrector with tau = 0.25 inflates the observed components' variance 1.4x ("hi") to 9.7x ("lo"), a
    predictor-only run collapses it) -- moments, emd and mmd of GPU-vs-oracle no worse than oracle-vs-oracle.
"""
import math

import pytest
import torch

from oracle import sda_oracle as O
from tests.util import assert_close, oracle_eps_from_module

pytestmark = pytest.mark.gpu
TOL = 1e-4
B, L, S, STEPS, TAU, GAMMA = 1024, 65, 3, 256, 0.25, 3e-2
FREQ = {'lo': (8, 0.05), 'hi': (1, 0.25)}


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _net(kind):
    from sda_amd.experiments.lorenz import make_global_score, make_local_score
    torch.manual_seed(50)
    return make_local_score() if kind == 'local' else make_global_score()


@pytest.mark.parametrize('kind', ['global', 'local'])
@pytest.mark.parametrize('freq', ['lo', 'hi'])
def test_lorenz_eval_batch_guided_evaluation_vs_oracle_on_a_slice(dev, kind, freq):
    from sda_amd import observe as Ob
    from sda_amd.score import GaussianScore, VPSDE
    step, std = FREQ[freq]
    net = _net(kind)
    eps_o = oracle_eps_from_module(net, 'local' if kind == 'local' else 'wrap1d')
    net.to(dev)
    torch.manual_seed(51)
    x = torch.randn(B, L, S)
    t = torch.tensor(0.6)
    A = lambda v: v[..., ::step, :1]
    y = torch.randn(A(x[0]).shape)                                      # one observation shared by the batch, as eval.py:47
    with torch.no_grad():
        out = net(x.to(dev), t.to(dev))
    rows = slice(480, 512)                                                # a slice in the middle of the batch
    assert_close(out[rows].cpu(), eps_o(x[rows], t), TOL, what=f'eps {kind}')
    ref = O.gaussian_score(eps_o, O.Schedule(), y, A, std, GAMMA, x[rows], t)
    for name, Aop in (('autograd through A', A), ('fused', Ob.Subsample((slice(None, None, step), slice(0, 1))))):
        gs = GaussianScore(y, A=Aop, std=std, sde=VPSDE(net, shape=()), gamma=GAMMA).to(dev)
        got = gs(x.to(dev), t.to(dev))
        assert torch.isfinite(got).all()
        assert_close(got[rows].cpu(), ref, TOL, what=f'guided {kind} {freq} ({name})')


def _oracle_samples(y, step, std, seed, corrections):
    sched = O.Schedule()
    eta = 1e-3
    A = lambda v: v[..., ::step, :1]
    eps = lambda x, t: x * (sched.sigma(t) / (1 + eta * eta))            # exact for N(0, I) data (bench.SyntheticScore, scale = 0)
    score = lambda x, t: O.gaussian_score(eps, sched, y, A, std, GAMMA, x, t)
    torch.manual_seed(seed)
    return O.sample(score, sched, torch.randn(B, L, S), 2, STEPS, corrections, TAU)


@pytest.mark.parametrize('freq', ['lo', 'hi'])
@pytest.mark.parametrize('graph', [False, True])
def test_lorenz_eval_statistical_end_to_end(dev, freq, graph):
    import bench
    from sda_amd import observe as Ob
    from sda_amd import utils as U
    from sda_amd.score import GaussianScore, VPSDE
    step, std = FREQ[freq]
    C = 1
    torch.manual_seed(60)
    y = torch.randn((L + step - 1) // step, 1) * math.sqrt(1 + std ** 2)          # y = A x + noise, x ~ N(0, I)
    ref_a = _oracle_samples(y, step, std, 61, C)
    ref_b = _oracle_samples(y, step, std, 62, C)

    net = _net('global').to(dev)
    score = bench.SyntheticScore(net, scale=0.0)                          # the net still runs (forward + VJP) at every evaluation
    inner = VPSDE(score, shape=())
    object.__setattr__(score, '_sched', inner)
    gs = GaussianScore(y, A=Ob.Subsample((slice(None, None, step), slice(0, 1))), std=std, sde=inner, gamma=GAMMA)
    sde = VPSDE(gs, shape=(L, S)).to(dev)
    sde.use_graph = graph
    torch.manual_seed(63)
    x = sde.sample((B,), steps=STEPS, corrections=C, tau=TAU).cpu()
    assert torch.isfinite(x).all()

    A = lambda v: v[..., ::step, :1]
    obs, oa, ob = A(x)[..., 0], A(ref_a)[..., 0], A(ref_b)[..., 0]       # (B, n_obs)
    n_obs = obs.shape[1]
    pm, pv = y[:, 0] / (1 + std ** 2), std ** 2 / (1 + std ** 2)
    zcrit = 4.5                                                            # two-sided 7e-6 per component; <= 65 components
    m, v = obs.mean(0), obs.var(0)
    # (a) analytic posterior mean, with the sample's own standard error (the corrector inflates the variance, not the mean)
    z = ((m - pm) / (v / B).sqrt()).abs().max().item()
    assert z < zcrit, f'{freq}: observed-component mean {z:.1f} sigma from the analytic posterior mean'
    # ... and the reference's own check (figures.ipynb#cell11): the residual spread is of the order of std
    spread, spread_o = (obs - y[:, 0]).std().item(), (oa - y[:, 0]).std().item()
    assert abs(spread / spread_o - 1) < 0.05, f'{freq}: (A(x) - y).std() = {spread:.4f}, oracle {spread_o:.4f} (std = {std})'
    # unobserved components: the prior N(0, 1) (mean over 1024 x 65 x 2 values; variance of the sample variance 2 / n)
    un = x[..., 1:]
    assert abs(un.mean().item()) < zcrit / math.sqrt(un.numel())
    # (b) against the oracle's loop: moments of the observed components within the two-sample error, variance ratio ~ 1
    z2 = ((m - oa.mean(0)) / ((v + oa.var(0)) / B).sqrt()).abs().max().item()
    assert z2 < zcrit, f'{freq}: observed-component mean {z2:.1f} sigma from the oracle sample set'
    ratio, ratio_ref = (v / oa.var(0)).mean().item(), (ob.var(0) / oa.var(0)).mean().item()
    # (the corrected samples are heavy-tailed on the observed components -- kurtosis ~ 5.5 at "lo", measured on the oracle's runs --,
    # so the variance of a sample variance is (kappa - 1) sigma^4 / B, not 2 sigma^4 / B; the inter-quartile range is the robust scale)
    dc = oa - oa.mean(0)
    kappa = max(3.0, ((dc ** 4).mean(0) / dc.var(0) ** 2).mean().item())
    tol_r = zcrit * math.sqrt(2 * (kappa - 1) / (B - 1) / n_obs) + 0.02
    iqr = lambda a: torch.quantile(a, 0.75, dim=0) - torch.quantile(a, 0.25, dim=0)
    iq = (iqr(obs) / iqr(oa)).mean().item()
    assert abs(iq - 1) < 0.10, f'{freq}: inter-quartile range of the observed components / oracle = {iq:.4f}'
    assert abs(ratio - 1) < tol_r, f'{freq}: observed variance / oracle variance = {ratio:.4f} (oracle vs oracle {ratio_ref:.4f}, tol {tol_r:.3f})'
    assert abs(un.var().item() / ref_a[..., 1:].var().item() - 1) < 0.02
    # W1 and MMD of the sample sets on the device (sda/utils.py:203-263), GPU-vs-oracle no worse than oracle-vs-oracle
    xa, xb, xg = ref_a.to(dev), ref_b.to(dev), x.to(dev)
    w_ref, w_got = float(U.emd(xa, xb)), float(U.emd(xg, xa))
    m_ref, m_got = float(U.mmd(xa, xb)), float(U.mmd(xg, xa))
    print(f'lorenz_eval {freq} graph={graph}: mean z {z:.2f} / {z2:.2f}, var ratio {ratio:.4f} (oracle pair {ratio_ref:.4f}), '
          f'emd {w_got:.4f} (oracle pair {w_ref:.4f}), mmd {m_got:.3e} (oracle pair {m_ref:.3e})')
    assert w_got < 1.05 * w_ref + 1e-3, f'{freq}: emd(GPU, oracle) {w_got:.4f} vs emd(oracle, oracle) {w_ref:.4f}'
    assert m_got < max(3 * abs(m_ref), 5e-3), f'{freq}: mmd(GPU, oracle) {m_got:.3e} vs mmd(oracle, oracle) {m_ref:.3e}'
