"""GPU parity on the BASELINE.json workloads themselves (configs[1], [3], [4]; configs[0] and [2] are covered in
test_gpu_net.py): the HIP path through the reference-mirroring API against the CPU oracle on seeded inputs, at the real
network sizes and resolutions but with as few trajectory windows as the oracle finishes in seconds, plus -- at the full
configs[3] shard -- the size-independent properties (finite, deterministic, group-streaming == one pass, VJP linear).

Shapes: experiments/kolmogorov/train.py:15-22 (K64 net), SURVEY.md section 8(d).  fp32, rtol 1e-4 (north_star)."""
import pytest
import torch

from oracle import sda_oracle as O
from tests.util import assert_close, oracle_eps_from_module, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
K64 = dict(window=5, embedding=64, hidden_channels=(96, 192, 384), hidden_blocks=(3, 3, 3))


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _subsample4(x):
    return x[..., ::4, ::4]


# ------------------------------------------------------------------------------------------------ configs[3]
@pytest.fixture(scope='module')
def k64_256(dev):
    from sda_amd.experiments.kolmogorov import make_score
    torch.manual_seed(30)
    net = make_score(size=256, **K64)
    eps_o = oracle_eps_from_module(net, 'mc2d')
    return net.to(dev), eps_o


def test_config3_k64_at_256_plain_and_guided(dev, k64_256):
    """K64 net at 256 x 256, one trajectory of L = 6 (two windows): the score and the Gaussian-guided score
    (A = every 4th pixel, as bench.py) vs the fp32 oracle; the fused-adjoint observation path gives the same numbers."""
    from sda_amd import observe as Ob
    from sda_amd.score import GaussianScore, VPSDE
    net, eps_o = k64_256
    torch.manual_seed(31)
    x = torch.randn(1, 6, 2, 256, 256)
    t = torch.tensor(0.37)
    with torch.no_grad():
        out = net(x.to(dev), t.to(dev))
        ref = eps_o(x, t)
    assert_close(out.cpu(), ref, TOL, what='eps 256^2')
    y = torch.randn(_subsample4(x).shape)
    ref_g = O.gaussian_score(eps_o, O.Schedule(), y, _subsample4, 0.1, 1e-2, x, t)
    gs = GaussianScore(y, A=_subsample4, std=0.1, sde=VPSDE(net, shape=())).to(dev)
    got = gs(x.to(dev), t.to(dev))
    assert_close(got.cpu(), ref_g, TOL, what='guided 256^2 (autograd through A)')
    gs2 = GaussianScore(y, A=Ob.Subsample.space(4), std=0.1, sde=VPSDE(net, shape=())).to(dev)
    got2 = gs2(x.to(dev), t.to(dev))
    assert_close(got2.cpu(), ref_g, TOL, what='guided 256^2 (fused adjoint)')


def test_config3_k64_at_256_vjp_vs_fp64_autograd(dev, k64_256):
    """J^T g through one 10 x 256 x 256 window vs torch autograd through the float64 oracle."""
    net, eps_o = k64_256
    torch.manual_seed(32)
    x = torch.randn(1, 5, 2, 256, 256)
    t = torch.tensor(0.61)
    g = torch.randn_like(x)
    xo = x.double().requires_grad_(True)
    eo = eps_o(xo, t.double(), torch.float64)
    ref, = torch.autograd.grad(eo, xo, g.double())
    xd = x.to(dev).requires_grad_(True)
    out = net(xd, t.to(dev))
    vjp, = torch.autograd.grad(out, xd, g.to(dev))
    assert_close(out.detach().cpu(), eo.detach(), TOL, what='eps (fp64 oracle)')
    assert_close(vjp.cpu(), ref, TOL, what='vjp (fp64 oracle)')


def test_config3_full_shard_properties(dev, k64_256):
    """The whole configs[3] per-GPU shard -- 16 trajectories x 64 x 2 x 256 x 256, 960 windows, guided -- goes through the
    group-streamed path (its activations exceed HBM).  Size-independent checks: finite; deterministic; the first
    trajectory equals the same trajectory evaluated alone (groups do not couple samples); the guidance VJP is linear in
    the observation residual (out(y1) + out(y2) - out(y0) relation of an affine map)."""
    from sda_amd import observe as Ob
    from sda_amd.score import GaussianScore, VPSDE
    net, _ = k64_256
    torch.manual_seed(33)
    B = 16
    x = torch.randn(B, 64, 2, 256, 256, device=dev)
    t = torch.tensor(0.5, device=dev)
    A = Ob.Subsample.space(4)
    ys = [torch.randn(B, 64, 2, 64, 64) for _ in range(2)]
    outs = []
    for y in ys:
        gs = GaussianScore(y, A=A, std=0.1, sde=VPSDE(net, shape=())).to(dev)
        outs.append(gs(x, t))
    assert torch.isfinite(outs[0]).all()
    gs = GaussianScore(ys[0], A=A, std=0.1, sde=VPSDE(net, shape=())).to(dev)
    again = gs(x, t)
    assert torch.equal(again, outs[0]), 'guided evaluation is not deterministic'
    solo = GaussianScore(ys[0][:1], A=A, std=0.1, sde=VPSDE(net, shape=())).to(dev)(x[:1], t)
    assert_close(outs[0][:1].cpu(), solo.cpu(), 1e-6, what='trajectory 0: streamed in a group vs alone')
    # out is affine in y: out(y) = a + M y  =>  out((y0 + y1)/2) = (out(y0) + out(y1))/2
    mid = GaussianScore((ys[0] + ys[1]) / 2, A=A, std=0.1, sde=VPSDE(net, shape=())).to(dev)(x, t)
    assert rel_err(mid, (outs[0] + outs[1]) / 2) < 1e-4


# ------------------------------------------------------------------------------------------------ configs[4]
def test_config4_qg_shaped_128_guided(dev):
    """QG-shaped workload: 4 state channels => Cin = 21 (20 + forcing), Cout = 20, K64-style net at
    128 x 128, guided with A = every 4th pixel, std 0.1.  One trajectory of L = 6 vs the oracle; VJP vs fp64 autograd."""
    from sda_amd.experiments.kolmogorov import LocalScoreUNet
    from sda_amd.score import GaussianScore, MCScoreNet, VPSDE
    from sda_amd.utils import ACTIVATIONS
    torch.manual_seed(40)
    net = MCScoreNet(4, order=2)
    net.kernel = LocalScoreUNet(channels=20, size=128, embedding=64, hidden_channels=(96, 192, 384),
                                hidden_blocks=(3, 3, 3), kernel_size=3, activation=ACTIVATIONS['SiLU'], spatial=2,
                                padding_mode='circular')
    eps_o = oracle_eps_from_module(net, 'mc2d')
    net.to(dev)
    x = torch.randn(1, 6, 4, 128, 128)
    t = torch.tensor(0.45)
    with torch.no_grad():
        out = net(x.to(dev), t.to(dev))
    assert_close(out.cpu(), eps_o(x, t), TOL, what='eps qg')
    y = torch.randn(_subsample4(x).shape)
    ref = O.gaussian_score(eps_o, O.Schedule(), y, _subsample4, 0.1, 1e-2, x, t)
    gs = GaussianScore(y, A=_subsample4, std=0.1, sde=VPSDE(net, shape=())).to(dev)
    assert_close(gs(x.to(dev), t.to(dev)).cpu(), ref, TOL, what='guided qg')
    # VJP of one window in float64
    x1 = torch.randn(1, 5, 4, 128, 128)
    g = torch.randn_like(x1)
    xo = x1.double().requires_grad_(True)
    ref_v, = torch.autograd.grad(eps_o(xo, t.double(), torch.float64), xo, g.double())
    xd = x1.to(dev).requires_grad_(True)
    got_v, = torch.autograd.grad(net(xd, t.to(dev)), xd, g.to(dev))
    assert_close(got_v.cpu(), ref_v, TOL, what='vjp qg')


def _full_shard_properties(dev, net, B, L, C, size, group):
    """The size-independent properties of test_config3_full_shard_properties on another configuration's whole per-GPU shard, with
    the group-streaming planner exercised explicitly: the shard fits HBM in one pass here, so the streamed form (`group`
    trajectories at a time, GaussianScore.group_size) must equal the one-pass form."""
    from sda_amd import observe as Ob
    from sda_amd.score import GaussianScore, VPSDE
    x = torch.randn(B, L, C, size, size, device=dev)
    t = torch.tensor(0.5, device=dev)
    A = Ob.Subsample.space(4)
    ys = [torch.randn(B, L, C, size // 4, size // 4) for _ in range(2)]

    def run(y, xs=x, group=None):
        gs = GaussianScore(y, A=A, std=0.1, sde=VPSDE(net, shape=())).to(dev)
        if group is not None:
            gs.group_size = group
        return gs(xs, t)
    outs = [run(y) for y in ys]
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(run(ys[0]), outs[0]), 'guided evaluation is not deterministic'
    errs = {}
    one_pass = run(ys[0], group=0)
    errs[f'streamed in groups of {group} vs one pass'] = rel_err(run(ys[0], group=group), one_pass)
    g2 = group + 1 if B % (group + 1) else group + 2                                 # a last group that is not full
    errs[f'ragged groups of {g2} vs one pass'] = rel_err(run(ys[0], group=g2), one_pass)
    errs['trajectory 0: in the batch vs alone'] = rel_err(outs[0][:1], run(ys[0][:1], xs=x[:1]))
    errs['last trajectory: in the batch vs alone'] = rel_err(outs[0][-1:], run(ys[0][-1:], xs=x[-1:]))
    # unguided score of the shard: windows of different trajectories do not mix (first / last trajectory alone)
    with torch.no_grad():
        e = net(x, t)
        errs['eps trajectory 0 vs alone'] = rel_err(e[:1], net(x[:1], t))
        errs['eps last trajectory vs alone'] = rel_err(e[-1:], net(x[-1:], t))
    print({k: f'{v:.2e}' for k, v in errs.items()})
    # (not bit-equality: the LayerNorm / direct-kernel variant and the persistent tile walk depend on how many windows a launch
    # carries, and each variant sums in its own order -- fp32 round-off through 42 convolutions and their VJPs)
    bad = {k: v for k, v in errs.items() if v > 1e-5}
    assert not bad, f'samples are coupled / the group planner changes results: {bad} (all: {errs})'
    mid = run((ys[0] + ys[1]) / 2)                                                   # affine in y
    assert rel_err(mid, (outs[0] + outs[1]) / 2) < 1e-4


def test_config2_full_shard_properties(dev):
    """BASELINE configs[2] whole: 32 trajectories x 32 x 2 x 64 x 64 (896 windows) through the reference Kolmogorov net, guided."""
    from sda_amd.experiments.kolmogorov import make_score
    torch.manual_seed(20)
    net = make_score(size=64, **K64).to(dev)
    torch.manual_seed(21)
    _full_shard_properties(dev, net, 32, 32, 2, 64, 8)


def test_config4_full_shard_properties(dev):
    """BASELINE configs[4]'s per-GPU shard whole: 8 trajectories x 32 x 4 x 128 x 128 (224 windows of 21 x 128 x 128), guided."""
    from sda_amd.experiments.kolmogorov import LocalScoreUNet
    from sda_amd.score import MCScoreNet
    from sda_amd.utils import ACTIVATIONS
    torch.manual_seed(41)
    net = MCScoreNet(4, order=2)
    net.kernel = LocalScoreUNet(channels=20, size=128, embedding=64, hidden_channels=(96, 192, 384),
                                hidden_blocks=(3, 3, 3), kernel_size=3, activation=ACTIVATIONS['SiLU'], spatial=2,
                                padding_mode='circular')
    net.to(dev)
    torch.manual_seed(42)
    _full_shard_properties(dev, net, 8, 32, 4, 128, 3)


# ------------------------------------------------------------------------------------------------ configs[1]
def test_config1_lorenz96_guided_eager_and_graph(dev):
    """Lorenz-96: 40 states, L = 128, batch 64, 1-D ScoreUNet (64,)/(3,), guided with the
    experiment's strided observation (experiments/lorenz/eval.py:75).  Score, guided score and teacher-forced PC steps
    with injected noise vs the oracle; the hipGraph-replayed step reproduces the eager one."""
    from sda_amd.experiments.lorenz import make_global_score
    from sda_amd.score import GaussianScore, VPSDE
    torch.manual_seed(10)
    net = make_global_score(channels=40)
    eps_o = oracle_eps_from_module(net, 'wrap1d')
    net.to(dev)
    B, L, S = 64, 128, 40
    x = torch.randn(B, L, S)
    t = torch.tensor(0.8)
    with torch.no_grad():
        out = net(x.to(dev), t.to(dev))
    assert_close(out.cpu(), eps_o(x, t), TOL, what='eps lorenz96')
    A = lambda v: v[..., ::8, :1]
    y = torch.randn(A(x).shape)
    gs = GaussianScore(y, A=A, std=0.5, sde=VPSDE(net, shape=()), gamma=3e-2).to(dev)
    ref = O.gaussian_score(eps_o, O.Schedule(), y, A, 0.5, 3e-2, x, t)
    assert_close(gs(x.to(dev), t.to(dev)).cpu(), ref, TOL, what='guided lorenz96')
    # three guided predictor-corrector steps with injected noise (teacher-forced: same z on both sides)
    steps, corr, tau = 3, 1, 0.25
    zs = torch.randn(steps * corr, B, L, S)
    sde = VPSDE(gs, shape=(L, S)).to(dev)
    sde.initial_noise = x
    sde.noise_source = lambda i, j: zs[i * corr + j]
    got = sde.sample((B,), steps=steps, corrections=corr, tau=tau)
    # the bound is what the reference arithmetic itself supports on this chain: the oracle run in fp32 against the oracle run
    # in fp64 (SURVEY 8c tier 3: "deviation from the fp32 reference <= the reference's own fp32-vs-fp64 deviation")
    score_o = lambda xx, tt: O.gaussian_score(eps_o, O.Schedule(), y, A, 0.5, 3e-2, xx, tt)
    ref_x = O.sample(score_o, O.Schedule(), x, 2, steps, corr, tau, noise=lambda i, j: zs[i * corr + j])
    y64, zs64 = y.double(), zs.double()
    score_64 = lambda xx, tt: O.gaussian_score(lambda a, b: eps_o(a, b, torch.float64), O.Schedule(), y64, A, 0.5, 3e-2, xx, tt)
    ref_64 = O.sample(score_64, O.Schedule(), x.double(), 2, steps, corr, tau, noise=lambda i, j: zs64[i * corr + j])
    own = rel_err(ref_x.double(), ref_64)
    err = rel_err(got.cpu().double(), ref_64)
    assert err <= max(TOL, 3 * own), (f'3 guided PC steps (6 evals deep): HIP path vs fp64 oracle {err:.2e}; the fp32 oracle itself is '
                                      f'{own:.2e} from the fp64 oracle on this chain (bound: max(1e-4, 3x that))')
    sde.noise_source = None
    # graph replay == eager on the device RNG stream
    outs = []
    for use_graph in (False, True):
        sde.initial_noise = x
        torch.manual_seed(11)
        sampler = sde.sampler((B,), steps=8, corrections=1, tau=0.25)
        if use_graph:
            sampler.capture()
        for _ in range(8):
            sampler.step()
        outs.append(sampler.result().clone())
    sde.initial_noise = None
    assert torch.isfinite(outs[0]).all()
    assert_close(outs[1].cpu(), outs[0].cpu(), 1e-5, what='graph vs eager')


def test_conv_sweep_large_images():
    """A bounded sample of tests/fuzz/conv_fuzz.py restricted to 128- and 256-pixel images (the configs[3]/[4] levels)."""
    import importlib.util
    import os
    import random
    spec = importlib.util.spec_from_file_location(
        'conv_fuzz', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fuzz', 'conv_fuzz.py'))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    from sda_amd import _lib
    _lib.load()
    rng = random.Random(256)
    failures = []
    for i in range(24):
        cfg, msg = fuzz.one_case(rng, torch.device('cuda:0'), 9000 + i, large=True)
        if msg:
            failures.append((cfg, msg))
    assert not failures, failures[:3]


# ------------------------------------------------------------------------------------------------ configs[0], end to end
@pytest.mark.parametrize('guided', [False, True])
def test_config0_lorenz63_256_steps_end_to_end(dev, guided):
    """BASELINE configs[0] as a whole: Lorenz-63 (3 states, L = 64, batch 1), the 1-D ScoreUNet of experiments/lorenz/utils.py:26-42,
    256 predictor steps from the same initial draw -- free-running, nothing teacher-forced -- against the oracle's sampling loop
    (sda/score.py:225-263), unguided and with Gaussian-likelihood guidance through the strided observation of
    experiments/lorenz/eval.py:75.  The network is random-init inside the synthetic estimator of SURVEY 8d (a raw random-init
    net overflows under guidance in the reference itself).  Bound: the reference arithmetic's own fp32-vs-fp64 deviation on
    this 256-evaluation chain (SURVEY 8c tier 3), floor 1e-4."""
    import bench
    from sda_amd.experiments.lorenz import make_global_score
    from sda_amd import observe as Ob
    from sda_amd.score import GaussianScore, VPSDE
    torch.manual_seed(30)
    net = make_global_score(channels=3)
    eps_net = oracle_eps_from_module(net, 'wrap1d')
    net.to(dev)
    steps, L, S = 256, 64, 3
    x1 = torch.randn(1, L, S)
    y = torch.randn(1, L // 8, 1)
    A = lambda v: v[..., ::8, :1]
    sched = O.Schedule()

    def oracle(dtype):
        def eps(xx, tt):
            mu, sg = sched.mu(tt), sched.sigma(tt)
            return xx * (sg / (mu * mu + sg * sg)) + 0.1 * eps_net(xx, tt, None if dtype == torch.float32 else dtype)
        score = (lambda xx, tt: O.gaussian_score(eps, sched, y.to(dtype), A, 0.5, 3e-2, xx, tt)) if guided else eps
        return O.sample(score, sched, x1.to(dtype), 2, steps, 0, 1.0)

    ref32, ref64 = oracle(torch.float32), oracle(torch.float64)
    assert torch.isfinite(ref64).all()
    score = bench.SyntheticScore(net)
    inner = VPSDE(score, shape=())
    mod = GaussianScore(y, A=Ob.Subsample((slice(None, None, 8), slice(0, 1))), std=0.5, sde=inner, gamma=3e-2) if guided else score
    sde = VPSDE(mod, shape=(L, S)).to(dev)
    sde.initial_noise = x1
    got = sde.sample((1,), steps=steps, corrections=0)
    own = rel_err(ref32.double(), ref64)
    err = rel_err(got.cpu().double(), ref64)
    print(f'configs[0] {steps}-step sample, guided={guided}: HIP vs fp64 oracle {err:.2e}, fp32 oracle vs fp64 oracle {own:.2e}')
    assert err <= max(TOL, 3 * own), (f'{steps}-step sample (guided={guided}): HIP path vs fp64 oracle {err:.2e}; the fp32 oracle itself is '
                                      f'{own:.2e} from the fp64 oracle (bound: max(1e-4, 3x that))')
