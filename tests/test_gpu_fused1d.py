"""GPU: the fused 1-D guided evaluation / predictor-corrector step (sda_amd/fused1d.py: sda_step1d_prologue, sda_net1d_fwd_fused,
sda_net1d_bwd_fused, sda_pc_correct_keyed) against the general path it replaces (same arithmetic in the same order: 1e-6) and the
oracle (sda/score.py:225-263, 375-396), over the shapes of experiments/lorenz (odd lengths, all three tile widths, per-sample and
shared observations, position / channel strides with offsets and stops, the synthetic estimator's affine form)."""
import pytest
import torch

from oracle import sda_oracle as O
from tests.util import assert_close, oracle_eps_from_module

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _build(dev, channels, affine, seed=70):
    import bench
    from sda_amd.experiments.lorenz import make_global_score
    from sda_amd.score import VPSDE
    torch.manual_seed(seed)
    net = make_global_score(channels=channels).to(dev)
    if affine:
        score = bench.SyntheticScore(net)
        inner = VPSDE(score, shape=())
        object.__setattr__(score, '_sched', inner)
    else:
        inner = VPSDE(net, shape=())
    return net, inner


CASES = [
    # B, L, C, slices, per-sample y, affine
    (1, 64, 3, (slice(None, None, 8), slice(0, 1)), False, True),          # BASELINE configs[0]: 32-column tiles
    (64, 128, 40, (slice(None, None, 8), slice(0, 1)), True, True),        # BASELINE configs[1]: 64-column tiles
    (5, 65, 3, (slice(None, None, 1), slice(0, 1)), False, False),         # eval.py "hi", odd length, bare network
    (9, 65, 3, (slice(3, 60, 5), slice(1, 3)), True, False),               # offsets, stops, two observed channels
    (4, 33, 7, (slice(0, None, 2),), False, True),                          # channel slice only (every position observed)
    (300, 17, 3, (slice(None, None, 4), slice(0, 3, 2)), True, True),       # sequences shorter than the halo, channel stride
]


@pytest.mark.parametrize('case', CASES)
def test_fused_guided_evaluation_equals_general_path_and_oracle(dev, case, monkeypatch):
    from sda_amd import fused1d, observe as Ob
    from sda_amd.score import GaussianScore
    B, L, C, sl, per_sample, affine = case
    net, inner = _build(dev, C, affine)
    torch.manual_seed(71)
    x = torch.randn(B, L, C)
    t = torch.tensor(0.37)
    A = Ob.Subsample(sl)
    oshape = A._osize(x.shape)
    y = torch.randn(oshape if per_sample else oshape[1:])
    gs = GaussianScore(y, A=A, std=0.3, sde=inner, gamma=3e-2).to(dev)
    xd, td = x.to(dev), t.to(dev)
    assert fused1d.plan(gs, xd, td, None) is not None, 'the fused plan declined a Lorenz-shaped job'
    got = gs(xd, td)
    got2 = gs(xd, td)
    assert got.data_ptr() != got2.data_ptr() and torch.equal(got, got2)
    monkeypatch.setattr(fused1d, 'ENABLED', False)
    ref = gs(xd, td)
    monkeypatch.setattr(fused1d, 'ENABLED', True)
    assert_close(got.cpu(), ref.cpu(), 1e-6, what='fused vs general path')
    # the oracle on (a slice of) the batch
    eps_net = oracle_eps_from_module(net, 'wrap1d')
    sched = O.Schedule()

    def eps_o(xx, tt):
        if not affine:
            return eps_net(xx, tt)
        mu, sg = sched.mu(tt), sched.sigma(tt)
        return xx * (sg / (mu * mu + sg * sg)) + 0.1 * eps_net(xx, tt)
    rows = slice(0, min(B, 8))
    Af = lambda v: v[(Ellipsis,) + tuple(sl)]
    ref_o = O.gaussian_score(eps_o, sched, y[rows] if per_sample else y, Af, 0.3, 3e-2, x[rows], t)
    assert_close(got[rows].cpu(), ref_o, TOL, what='fused vs oracle')


@pytest.mark.parametrize('B,L,C,corr', [(1, 64, 3, 1), (64, 128, 40, 1), (33, 65, 3, 2), (7, 65, 3, 0)])
@pytest.mark.parametrize('noise', ['keyed', 'torch'])
def test_fused_pc_steps_equal_general_path_eager_and_graph(dev, B, L, C, corr, noise, monkeypatch):
    """Free-running predictor-corrector steps (sda/score.py:250-261): fused eager == fused graph replay == the general path, with
    the row-keyed in-kernel noise (sda_pc_correct_keyed generates the z of sda_randn_rows) and with the device RNG."""
    from sda_amd import fused1d, observe as Ob, parallel
    from sda_amd.score import GaussianScore, VPSDE
    net, inner = _build(dev, C, True, seed=72)
    torch.manual_seed(73)
    x1 = torch.randn(B, L, C)
    A = Ob.Subsample((slice(None, None, 8), slice(0, 1)))
    y = torch.randn(A._osize(x1.shape))
    gs = GaussianScore(y, A=A, std=0.2, sde=inner, gamma=3e-2)
    sde = VPSDE(gs, shape=(L, C)).to(dev)
    steps_run, steps = 6, 50

    def run(fused, graph):
        monkeypatch.setattr(fused1d, 'ENABLED', fused)
        sde.initial_noise = x1
        sde.noise_source = parallel.KeyedNoise((5, 5 + B), (L, C), 9, corr, dev) if noise == 'keyed' and corr else None
        torch.manual_seed(74)
        sampler = sde.sampler((B,), steps=steps, corrections=corr, tau=0.25)
        assert (sampler._fused is not None) == fused
        if graph:
            sampler.capture()
        for _ in range(steps_run):
            sampler.step()
        torch.cuda.synchronize()
        sde.initial_noise, sde.noise_source = None, None
        return sampler.result().clone()

    base = run(False, False)
    assert torch.isfinite(base).all()
    fe = run(True, False)
    assert_close(fe.cpu(), base.cpu(), 2e-5, what=f'{steps_run} fused steps vs the general path')
    fg = run(True, True)
    assert_close(fg.cpu(), fe.cpu(), 1e-6, what='fused graph replay vs fused eager')
    monkeypatch.setattr(fused1d, 'ENABLED', True)


def test_fused_step_prologue_and_keyed_correction_are_bit_identical_to_the_kernels_they_replace(dev):
    """sda_step1d_prologue == sda_vp_schedule + sda_time_embed + sda_linear_small; sda_pc_correct_keyed == sda_randn_rows +
    sda_pc_correct (same Philox counters, same update)."""
    from sda_amd import fused1d, observe as Ob, ops
    from sda_amd.score import GaussianScore
    net, inner = _build(dev, 3, False, seed=75)
    A = Ob.Subsample((slice(None, None, 8), slice(0, 1)))
    x = torch.randn(4, 65, 3, device=dev)
    gs = GaussianScore(torch.randn(9, 1), A=A, std=0.1, sde=inner).to(dev)
    t = torch.tensor(0.81, device=dev)
    fz = fused1d.plan(gs, x, t, None)
    table = torch.tensor([[0.81, 0.80, 0.97, 0.013, 0.4], [0.5, 0.49, 0.96, 0.02, 0.3]], device=dev)
    istep = torch.ones(1, device=dev, dtype=torch.int64)
    fz.prologue_step(table, istep)
    torch.cuda.synchronize()
    assert istep.item() == 2 and fz.step_i.item() == 1
    score = net.score
    for k, tv in enumerate((0.5, 0.49)):
        tt = torch.tensor(tv, device=dev)
        mu, sg = inner.mu_sigma(tt)
        assert fz.coef[2 * k].item() == mu.item() and fz.coef[2 * k + 1].item() == sg.item()
        mod = score.network.engine().modulation(score.embedding(tt.reshape(1)))
        assert torch.equal(fz.mod[k], mod[0])
    assert torch.equal(fz.coef[4:7], table[1, 2:5]) and torch.equal(fz.coef[7:9], table[1, :2])
    # keyed correction
    B, per = 6, 65 * 3
    xa = torch.randn(B, 65, 3, device=dev)
    xb = xa.clone()
    eps = torch.randn_like(xa)
    partial = torch.rand(B, 3, device=dev) * 50 + 1
    coef = torch.tensor([0.37], device=dev)
    draw = torch.tensor([11], device=dev, dtype=torch.int64)
    z = ops.randn_rows(torch.empty_like(xa), 1234567, 40, draw_dev=draw, draw_mul=2, draw_add=1)
    ops.pc_correct(xa, eps, z, B, partial, 0.25, 0.0, coef_dev=coef, nchunk=3)
    ops.pc_correct_keyed(xb, eps, B, partial, 3, 0.25, coef, 1234567, 40, draw, 2, 1)
    assert torch.equal(xa, xb)


@pytest.mark.parametrize('B,L,C,blocks', [(300, 65, 3, 3), (300, 40, 3, 3), (520, 20, 5, 2), (1024, 65, 3, 3), (270, 80, 40, 1), (300, 12, 3, 3)])
def test_whole_sequence_tiles_forward_and_vjp_vs_oracle(dev, B, L, C, blocks):
    """Zero-padded sequences that fit one tile and come in numbers that fill the chip run as WHOLE-sequence tiles (no halo, csrc/net1d.hip:
    NF = ceil(L / 16) = 5, 3, 2, 5, 5, 2 here): the network and its input VJP against the float64 oracle (sda/nn.py:184-206)."""
    from sda_amd.experiments.lorenz import make_global_score
    torch.manual_seed(80 + L)
    net = make_global_score(channels=C, hidden_blocks=(blocks,)).to(dev)
    eps_o = oracle_eps_from_module(net, 'wrap1d')
    x = torch.randn(B, L, C)
    t = torch.tensor(0.55)
    g = torch.randn(B, L, C)
    rows = slice(B - 6, B)
    xo = x[rows].double().requires_grad_(True)
    eo = eps_o(xo, t.double(), torch.float64)
    ref_v, = torch.autograd.grad(eo, xo, g[rows].double())
    xd = x.to(dev).requires_grad_(True)
    out = net(xd, t.to(dev))
    vjp, = torch.autograd.grad(out, xd, g.to(dev))
    assert_close(out[rows].detach().cpu(), eo.detach(), TOL, what='eps (whole-sequence tiles)')
    assert_close(vjp[rows].cpu(), ref_v, TOL, what='vjp (whole-sequence tiles)')
    # rows do not couple: the first rows alone (a batch that takes the halo-tiled kernels) give the same values
    xs = x[:3].to(dev).requires_grad_(True)
    o3 = net(xs, t.to(dev))
    v3, = torch.autograd.grad(o3, xs, g[:3].to(dev))
    assert_close(out[:3].detach().cpu(), o3.detach().cpu(), 1e-5, what='whole-sequence vs halo tiles (eps)')
    assert_close(vjp[:3].cpu(), v3.cpu(), 1e-5, what='whole-sequence vs halo tiles (vjp)')


def _build_local(dev, features, window, affine, seed=90):
    import bench
    from sda_amd.experiments.lorenz import make_local_score
    from sda_amd.score import VPSDE
    torch.manual_seed(seed)
    net = make_local_score(window=window, features=features).to(dev)
    if affine:
        score = bench.SyntheticScore(net)
        inner = VPSDE(score, shape=())
        object.__setattr__(score, '_sched', inner)
    else:
        inner = VPSDE(net, shape=())
    return net, inner


LOCAL_CASES = [
    # B, L, C, window, slices, per-sample y, affine
    (5, 65, 3, 5, (slice(None, None, 8), slice(0, 1)), False, True),       # eval.py "lo" (experiments/lorenz/eval.py:50-53)
    (1100, 65, 3, 5, (slice(None, None, 1), slice(0, 1)), False, False),   # eval.py "hi" at its batch size, bare network
    (9, 17, 3, 5, (slice(3, 15, 5), slice(1, 3)), True, False),            # offsets, stops, two observed channels
    (1, 5, 3, 5, (slice(0, None, 2),), False, True),                        # one window per trajectory, channel slice only
    (70, 30, 5, 3, (slice(None, None, 4), slice(0, 5, 2)), True, True),     # window 3 of five states
    (33, 12, 2, 5, (slice(None, None, 3), slice(0, 1)), True, False),       # two states
]


@pytest.mark.parametrize('case', LOCAL_CASES)
def test_fused_local_evaluation_equals_general_path_and_oracle(dev, case, monkeypatch):
    """The fused evaluation of a LOCAL score network (MCScoreNet over a ScoreNet, experiments/lorenz/utils.py:45-59: sda_mlp_fwd_win,
    sda_mlp_bwd_win, sda_mc_finish) against the general path (unfold + cat + whole-MLP kernels + fold + the guidance kernels) and the oracle."""
    from sda_amd import fused1d, observe as Ob
    from sda_amd.score import GaussianScore
    B, L, C, window, sl, per_sample, affine = case
    net, inner = _build_local(dev, C, window, affine)
    torch.manual_seed(91)
    x = torch.randn(B, L, C)
    t = torch.tensor(0.41)
    A = Ob.Subsample(sl)
    oshape = A._osize(x.shape)
    y = torch.randn(oshape if per_sample else oshape[1:])
    gs = GaussianScore(y, A=A, std=0.3, sde=inner, gamma=3e-2).to(dev)
    xd, td = x.to(dev), t.to(dev)
    fz = fused1d.plan(gs, xd, td, None)
    assert isinstance(fz, fused1d.FusedLocal), 'the fused plan declined a local Lorenz-shaped job'
    got = gs(xd, td)
    assert torch.equal(got, gs(xd, td))
    monkeypatch.setattr(fused1d, 'ENABLED', False)
    ref = gs(xd, td)
    monkeypatch.setattr(fused1d, 'ENABLED', True)
    assert_close(got.cpu(), ref.cpu(), 1e-5, what='fused vs general path (local net)')
    eps_net = oracle_eps_from_module(net, 'local')
    sched = O.Schedule()

    def eps_o(xx, tt):
        if not affine:
            return eps_net(xx, tt)
        mu, sg = sched.mu(tt), sched.sigma(tt)
        return xx * (sg / (mu * mu + sg * sg)) + 0.1 * eps_net(xx, tt)
    rows = slice(max(0, B - 6), B)
    Af = lambda v: v[(Ellipsis,) + tuple(sl)]
    ref_o = O.gaussian_score(eps_o, sched, y[rows] if per_sample else y, Af, 0.3, 3e-2, x[rows], t)
    assert_close(got[rows].cpu(), ref_o, TOL, what='fused vs oracle (local net)')


@pytest.mark.parametrize('B,L,corr', [(1, 65, 1), (300, 65, 2), (40, 9, 0)])
@pytest.mark.parametrize('noise', ['keyed', 'torch'])
def test_fused_local_pc_steps_equal_general_path_eager_and_graph(dev, B, L, corr, noise, monkeypatch):
    from sda_amd import fused1d, observe as Ob, parallel
    from sda_amd.score import GaussianScore, VPSDE
    C = 3
    net, inner = _build_local(dev, C, 5, True, seed=92)
    torch.manual_seed(93)
    x1 = torch.randn(B, L, C)
    A = Ob.Subsample((slice(None, None, 8), slice(0, 1)))
    y = torch.randn(A._osize(x1.shape)[1:])
    gs = GaussianScore(y, A=A, std=0.2, sde=inner, gamma=3e-2)
    sde = VPSDE(gs, shape=(L, C)).to(dev)

    def run(fused, graph):
        monkeypatch.setattr(fused1d, 'ENABLED', fused)
        sde.initial_noise = x1
        sde.noise_source = parallel.KeyedNoise((5, 5 + B), (L, C), 9, corr, dev) if noise == 'keyed' and corr else None
        torch.manual_seed(94)
        sampler = sde.sampler((B,), steps=50, corrections=corr, tau=0.25)
        assert isinstance(sampler._fused, fused1d.FusedLocal) == fused
        if graph:
            sampler.capture()
        for _ in range(6):
            sampler.step()
        torch.cuda.synchronize()
        sde.initial_noise, sde.noise_source = None, None
        return sampler.result().clone()

    base = run(False, False)
    assert torch.isfinite(base).all()
    fe = run(True, False)
    assert_close(fe.cpu(), base.cpu(), 5e-5, what='6 fused steps vs the general path (local net)')
    fg = run(True, True)
    assert_close(fg.cpu(), fe.cpu(), 1e-6, what='fused graph replay vs fused eager (local net)')
    monkeypatch.setattr(fused1d, 'ENABLED', True)
