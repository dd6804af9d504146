"""CPU: the Python orchestration of sda_amd (engine forward / VJP sequencing, autograd glue, guidance chain rule,
PC loop, reference-compatible API) executed on CPU tensors through the TEST-ONLY shim of tests/cpu_shim.py
(convolutions = host replay of the gfx950 tile algorithm), checked against the fixtures the reference's own code
produced.  The device kernels themselves are covered by the -m gpu tests."""
import pytest
import torch
import torch.nn as nn

from tests import cpu_shim
from tests.util import (assert_close, build_mcscore2d_tiny, build_unet1d_tiny, build_unet1d_two_level, load_golden,
                        oracle_eps_from_module)

TOL = 2e-5


@pytest.fixture(autouse=True)
def shim(monkeypatch):
    cpu_shim.install(monkeypatch)
    # isinstance(x.is_cuda) gates in score.py
    yield


def _A(x):
    return x[..., ::2, :, ::2, ::2]


def test_unet1d_wrapper_golden():
    g, grp = load_golden('unet1d_tiny')
    net = build_unet1d_tiny()
    net.load_state_dict(grp['sd'])
    with torch.no_grad():
        out = net(g['x'], g['t'])
    assert_close(out, g['out'], TOL)


def test_unet1d_two_level_per_sample_time_golden():
    g, grp = load_golden('unet1d_two_level')
    net = build_unet1d_two_level()
    net.load_state_dict(grp['sd'])
    with torch.no_grad():
        out = net(g['x'], g['t'])
    assert_close(out, g['out'], TOL)


def test_mcscore2d_fused_golden_and_vjp():
    g, grp = load_golden('mcscore2d_tiny')
    net = build_mcscore2d_tiny()
    net.load_state_dict(grp['sd'])
    with torch.no_grad():
        out = net(g['x'], g['t'])
    assert_close(out, g['out'], TOL)
    # VJP against autograd through the oracle
    eps_o = oracle_eps_from_module(net, 'mc2d')
    torch.manual_seed(0)
    gg = torch.randn_like(g['x'])
    xo = g['x'].clone().requires_grad_(True)
    ref, = torch.autograd.grad(eps_o(xo, g['t']), xo, gg)
    xs = g['x'].clone().requires_grad_(True)
    got, = torch.autograd.grad(net(xs, g['t']), xs, gg)
    assert_close(got, ref, 5e-5)


def test_guided_and_dps_golden():
    from sda_amd.score import DPSGaussianScore, GaussianScore, VPSDE
    g, grp = load_golden('mcscore2d_tiny')
    net = build_mcscore2d_tiny()
    net.load_state_dict(grp['sd'])
    inner = VPSDE(net, shape=())
    gs = GaussianScore(g['y_obs'], A=_A, std=0.5, sde=inner, gamma=1e-2)
    assert_close(gs(g['x'], g['t_guided']), g['guided'], 5e-5)
    assert_close(gs.log_p_grad(g['x'], g['t_guided']), g['grad_logp_ref'], 1e-4)    # the reference's autograd.grad output
    # streamed in groups of whole trajectories (what a batch too large for HBM gets): same result, shared or per-row y
    gs.group_size = 1
    assert_close(gs(g['x'], g['t_guided']), g['guided'], 5e-5)
    x3 = torch.cat((g['x'], g['x'][:1] * 0.5), 0)
    y3 = _A(x3) + 0.1
    per_row = GaussianScore(y3, A=_A, std=0.5, sde=inner, gamma=1e-2)
    per_row.group_size = 0
    whole = per_row(x3, g['t_guided'])
    per_row.group_size = 2
    assert_close(per_row(x3, g['t_guided']), whole, 1e-6)
    dps = DPSGaussianScore(g['y_obs'], A=_A, sde=inner, zeta=1.0)
    assert_close(dps(g['x'], g['t_guided']), g['dps'], 5e-5)
    gsd = GaussianScore(g['y_obs'], A=_A, std=0.5, sde=inner, gamma=1e-2, detach=True)
    assert gsd(g['x'], g['t_guided']).shape == g['guided'].shape


def test_guided_pc_steps_golden():
    from sda_amd.score import GaussianScore, VPSDE
    g, grp = load_golden('mcscore2d_tiny')
    net = build_mcscore2d_tiny()
    net.load_state_dict(grp['sd'])
    gs = GaussianScore(g['y_obs'], A=_A, std=0.5, sde=VPSDE(net, shape=()), gamma=1e-2)
    sde = VPSDE(gs, shape=(5, 2, 8, 8))
    steps, corr, tau = int(g['pc_args'][0]), int(g['pc_args'][1]), float(g['pc_args'][2])
    zs = g['pc_noise']
    sde.initial_noise = g['pc_x_init']
    sde.noise_source = lambda i, j: zs[i * corr + j]
    x = sde.sample((2,), steps=steps, corrections=corr, tau=tau)
    assert_close(x, g['pc_x_final'], 1e-4)


def test_unguided_sampling_golden():
    from sda_amd.score import VPSDE
    g, _ = load_golden('sample_unguided_lorenz')
    _, grp = load_golden('unet1d_tiny')
    net = build_unet1d_tiny()
    net.load_state_dict(grp['sd'])
    sde = VPSDE(net, shape=(16, 3))
    zs = g['noise']
    sde.initial_noise = g['x_init']
    sde.noise_source = lambda i, j: zs[i * 2 + j]
    x = sde.sample((3,), steps=8, corrections=2, tau=0.25)
    assert_close(x, g['x_final'], 1e-4)


def test_chunked_recompute_path(monkeypatch):
    from sda_amd import engine as E
    g, grp = load_golden('mcscore2d_tiny')
    net = build_mcscore2d_tiny()
    net.load_state_dict(grp['sd'])
    torch.manual_seed(1)
    gg = torch.randn_like(g['x'])
    xs = g['x'].clone().requires_grad_(True)
    out = net(xs, g['t'])
    v, = torch.autograd.grad(out, xs, gg)
    monkeypatch.setattr(E.UNetEngine, 'chunk_size', lambda self, n, hs, ws, save, device, fraction=None: min(n, 2))
    xs2 = g['x'].clone().requires_grad_(True)
    out2 = net(xs2, g['t'])
    v2, = torch.autograd.grad(out2, xs2, gg)
    assert torch.equal(out, out2) and torch.equal(v, v2)


def test_generic_kernel_path_user_subclass():
    from sda_amd.score import MCScoreNet, ScoreUNet
    g, grp = load_golden('mcscore2d_tiny')

    class UserLocal(ScoreUNet):
        def __init__(self, channels, size, **kw):
            super().__init__(channels, 1, **kw)
            self.register_buffer('forcing', torch.zeros(1, size, size))

        def forward(self, x, t, c=None):
            return super().forward(x, t, self.forcing)

    net = MCScoreNet(2, order=1)
    net.kernel = UserLocal(6, 8, embedding=8, hidden_channels=(4, 8), hidden_blocks=(1, 1), kernel_size=3,
                           activation=nn.SiLU, spatial=2, padding_mode='circular')
    net.load_state_dict(grp['sd'])
    with torch.no_grad():
        assert_close(net(g['x'], g['t']), g['out'], TOL)


def test_scorenet_local_golden_and_vjp():
    """Lorenz local kernel: MCScoreNet(features=3, order=2) over a ScoreNet/ResMLP (score.py:38-63, nn.py:31-71)."""
    from oracle import sda_oracle as O
    from sda_amd.score import MCScoreNet
    g, grp = load_golden('scorenet_local_tiny')
    net = MCScoreNet(features=3, order=2, embedding=8, hidden_features=[16] * 2, activation=nn.SiLU)
    net.load_state_dict(grp['sd'])
    with torch.no_grad():
        out = net(g['x'], g['t'])
    assert_close(out, g['out'], TOL)
    cfg = O.ResMLPConfig(15 + 8, 15, (16, 16), 'SiLU')
    sd = grp['sd']
    eps_o = lambda x, t: O.mc_score_net(lambda a, b, c: O.score_net(sd, 'kernel.', cfg, a, b, c), 2, x, t)
    torch.manual_seed(0)
    gg = torch.randn_like(g['x'])
    xo = g['x'].clone().requires_grad_(True)
    ref, = torch.autograd.grad(eps_o(xo, g['t']), xo, gg)
    xs = g['x'].clone().requires_grad_(True)
    got, = torch.autograd.grad(net(xs, g['t']), xs, gg)
    assert_close(got, ref, 5e-5)


def test_stride2_vjp_parity_split_equals_zero_insertion(monkeypatch):
    """The level heads' VJP as four (two, 1-D) parity-class convolutions == one convolution of the zero-inserted tensor,
    and the split path really is the one taken (launches with an explicit pad and an interleaved output view)."""
    from sda_amd import engine as E
    from sda_amd import ops
    seen = []
    real = ops.conv_igemm

    def spy(desc):
        seen.append((desc.explicit_pad, desc.kh, desc.kw, desc.zins_h, desc.zins_w, desc.out_sx))
        return real(desc)
    monkeypatch.setattr(ops, 'conv_igemm', spy)
    g, grp = load_golden('mcscore2d_tiny')
    net2 = build_mcscore2d_tiny()
    net2.load_state_dict(grp['sd'])
    g1, grp1 = load_golden('unet1d_two_level')
    net1 = build_unet1d_two_level()
    net1.load_state_dict(grp1['sd'])
    for net, x, t, want in ((net2, g['x'], g['t'], {(1, 1), (1, 2), (2, 1), (2, 2)}), (net1, g1['x'], g1['t'], {(1, 1), (1, 2)})):
        torch.manual_seed(4)
        gg = torch.randn_like(x)
        grads = {}
        for split in (True, False):
            monkeypatch.setattr(E, 'PARITY_SPLIT', split)
            seen.clear()
            xs = x.clone().requires_grad_(True)
            grads[split], = torch.autograd.grad(net(xs, t), xs, gg)
            if split:
                assert {(kh, kw) for ep, kh, kw, zh, zw, sx in seen if ep} == want
                assert all(sx == 2 for ep, kh, kw, zh, zw, sx in seen if ep)
                assert not any(zh > 1 or zw > 1 for ep, kh, kw, zh, zw, sx in seen)
            else:
                assert any(zh > 1 or zw > 1 for ep, kh, kw, zh, zw, sx in seen) and not any(ep for ep, *_ in seen)
        assert_close(grads[True], grads[False], 1e-6)


class _SubAdj:
    """test-local linear observation with a hand-written adjoint (takes GaussianScore's gauss_cotangent fast path)."""

    def __call__(self, x):
        return _A(x)

    def adjoint(self, r, x_shape):
        g = torch.zeros(x_shape)
        g[..., ::2, :, ::2, ::2] = r
        return g


def test_gaussian_score_reads_live_std_gamma():
    """std / gamma are buffers (state_dict entries); the fused paths must follow load_state_dict, assignment and
    in-place writes exactly as the general path does (reference score.py:387 reads self.std / self.gamma at call time)."""
    from sda_amd.score import GaussianScore, VPSDE
    g, grp = load_golden('mcscore2d_tiny')
    net = build_mcscore2d_tiny()
    net.load_state_dict(grp['sd'])
    inner = VPSDE(net, shape=())
    x, t = g['x'], g['t_guided']

    def fresh(std, gamma, A):
        return GaussianScore(g['y_obs'], A=A, std=std, sde=inner, gamma=gamma)(x, t)

    for A in (_A, _SubAdj()):
        want = fresh(0.2, 5e-2, A)
        gs = GaussianScore(g['y_obs'], A=A, std=0.5, sde=inner, gamma=1e-2)
        base = gs(x, t)                                              # (fills the scalar cache with the constructor values)
        assert (base - want).abs().max() > 1e-4 * want.abs().max()  # the two settings do differ
        gs.std.fill_(0.2); gs.gamma.fill_(5e-2)                      # in place
        assert gs._scalars == pytest.approx((0.2, 5e-2))
        assert_close(gs(x, t), want, 1e-6)
        gs2 = GaussianScore(g['y_obs'], A=A, std=0.5, sde=inner, gamma=1e-2)
        gs2(x, t)
        gs2.load_state_dict(gs.state_dict())                         # checkpoint round trip
        assert_close(gs2(x, t), want, 1e-6)
        gs3 = GaussianScore(g['y_obs'], A=A, std=0.5, sde=inner, gamma=1e-2)
        gs3(x, t)
        gs3.std = torch.tensor(0.2); gs3.gamma = torch.tensor(5e-2)  # buffer reassignment
        assert_close(gs3(x, t), want, 1e-6)
    # non-scalar std: no scalar fast path, general path as before
    gv = GaussianScore(g['y_obs'], A=_A, std=torch.full(g['y_obs'].shape[-3:], 0.2), sde=inner, gamma=5e-2)
    assert gv._scalars is None
    assert_close(gv(x, t), fresh(0.2, 5e-2, _A), 1e-5)


@pytest.mark.parametrize('case', ['a', 'b'])
def test_unet3d_orchestration_golden(case):
    """``UNet(spatial=3)``: engine3d's sequencing (heads, modulated blocks, LayerNorm -> up-sample -> conv tails, skips) and its
    hand-written input VJP against the fixture the reference's own Conv3d U-Net produced (launches = torch conv3d stand-ins)."""
    from sda_amd.score import ScoreUNet
    g, grp = load_golden('unet3d_tiny')
    if case == 'a':
        net = ScoreUNet(2, context=1, embedding=8, hidden_channels=(4, 8), hidden_blocks=(1, 1), kernel_size=3,
                        activation=nn.SiLU, spatial=3, padding_mode='circular')
    else:
        net = ScoreUNet(3, embedding=8, hidden_channels=(5, 20), hidden_blocks=(1, 2), kernel_size=(1, 3, 3), stride=(1, 2, 2),
                        activation=nn.ELU, spatial=3)
    net.load_state_dict(grp['sd_' + case])
    x = g['x_' + case].clone().requires_grad_(True)
    out = net(x, g['t_' + case], g.get('c_' + case))
    assert_close(out, g['out_' + case], TOL)
    gx, = torch.autograd.grad((out * g['cot_' + case]).sum(), x)
    assert_close(gx, g['gx_' + case], TOL)


def test_denoising_loss_is_a_value_only():
    """``VPSDE.loss`` (sda/score.py:265-276): the reference's validation quantity under no_grad; a call that would need parameter
    gradients is refused (the networks form input gradients only)."""
    from sda_amd.score import VPSDE
    g, grp = load_golden('unet1d_two_level')
    net = build_unet1d_two_level()
    net.load_state_dict(grp['sd'])
    sde = VPSDE(net, shape=(3, 20))
    x = g['x']
    with pytest.raises(NotImplementedError, match='parameter gradients'):
        sde.loss(x)
    torch.manual_seed(3)
    with torch.no_grad():
        got = sde.loss(x)
    torch.manual_seed(3)
    t = torch.rand(x.shape[0])
    eps = torch.randn_like(x)
    from oracle import sda_oracle as O
    sched = O.Schedule()
    tb = t.double().reshape(-1, 1, 1)
    xt = sched.mu(tb) * x.double() + sched.sigma(tb) * eps.double()
    cfg = O.UNetConfig(3, 3, 8, (8, 16), (1, 2), 3, 2, 'SiLU', 1, 'zeros')
    sd = {k: v.double() for k, v in grp['sd'].items()}
    want = (O.score_unet(sd, '', cfg, xt, t.double()) - eps.double()).square().mean()
    assert abs(got.item() - want.item()) <= 2e-5 * abs(want.item())
    for p in net.parameters():
        p.requires_grad_(False)
    assert torch.isfinite(sde.loss(x))                   # frozen parameters: allowed with grad mode on
