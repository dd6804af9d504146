"""GPU: ``UNet(spatial=3)`` / ``ScoreUNet(..., spatial=3)`` (sda/nn.py:114-118, 148-206; score.py:66-93) on the general 3-D kernel
(csrc/conv3d.hip, sda_amd/engine3d.py) -- against fixtures the reference's own code produced (forward and autograd input gradient),
each index map of the kernel against torch's conv3d in float64, and a guided evaluation against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sda_oracle as O
from tests.util import assert_close, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _net(case, dev):
    from sda_amd.score import ScoreUNet
    if case == 'a':
        net = ScoreUNet(2, context=1, embedding=8, hidden_channels=(4, 8), hidden_blocks=(1, 1), kernel_size=3,
                        activation=torch.nn.SiLU, spatial=3, padding_mode='circular')
    else:
        net = ScoreUNet(3, embedding=8, hidden_channels=(5, 20), hidden_blocks=(1, 2), kernel_size=(1, 3, 3), stride=(1, 2, 2),
                        activation=torch.nn.ELU, spatial=3)
    g, grp = load_golden('unet3d_tiny')
    net.load_state_dict(grp['sd_' + case])
    return net.to(dev), g


@pytest.mark.parametrize('case', ['a', 'b'])
def test_unet3d_matches_reference_fixture(dev, case):
    net, g = _net(case, dev)
    x = g['x_' + case].to(dev).requires_grad_(True)
    c = g['c_' + case].to(dev) if case == 'a' else None
    out = net(x, g['t_' + case].to(dev), c)
    assert out.shape == x.shape
    assert_close(out, g['out_' + case], 1e-4, what='forward')
    gx, = torch.autograd.grad((out * g['cot_' + case].to(dev)).sum(), x)       # hand-written VJP (engine3d.backward_all)
    assert_close(gx, g['gx_' + case], 1e-4, what='input gradient')
    with torch.no_grad():                                                     # nothing saved, same numbers
        assert torch.equal(net(x.detach(), g['t_' + case].to(dev), c), out.detach())


def _ref_conv(x, w, b, stride, circular, up):
    """float64 torch: nearest up-sampling, padding k // 2 in the given mode, stride."""
    for ax, u in enumerate(up):
        x = x.repeat_interleave(u, dim=2 + ax)
    pads = [k // 2 for k in w.shape[2:]]
    if circular:
        flat = []
        for p in reversed(pads):
            flat += [p, p]
        return F.conv3d(F.pad(x, flat, mode='circular'), w, b, stride=stride)
    return F.conv3d(x, w, b, stride=stride, padding=pads)


@pytest.mark.parametrize('cin,cout,size,k,stride,up,circular', [
    (3, 5, (4, 6, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    (7, 70, (3, 5, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1), False),          # odd sizes, more than one cout tile, ragged cin quad
    (6, 18, (4, 6, 8), (3, 3, 3), (2, 2, 2), (1, 1, 1), True),           # stride-2 head
    (6, 18, (5, 6, 7), (3, 3, 3), (2, 2, 2), (1, 1, 1), False),          # ... zero padding, odd sizes
    (9, 4, (2, 3, 4), (3, 3, 3), (1, 1, 1), (2, 2, 2), True),            # up-sampled tail
    (9, 4, (2, 3, 4), (1, 3, 5), (1, 1, 1), (1, 2, 2), False),           # anisotropic kernel and scale
    (5, 33, (4, 4, 6), (3, 1, 3), (1, 2, 2), (1, 1, 1), False),
])
def test_conv3d_forward_and_vjp_against_float64(dev, cin, cout, size, k, stride, up, circular):
    from sda_amd.engine3d import _Conv3d
    torch.manual_seed(cin * 100 + cout)
    conv = torch.nn.Conv3d(cin, cout, k, stride=stride, padding=[kk // 2 for kk in k],
                           padding_mode='circular' if circular else 'zeros').to(dev)
    x = torch.randn(2, cin, *size, device=dev)
    op = _Conv3d(conv)
    xd = x.double().cpu().requires_grad_(True)
    ref = _ref_conv(xd, conv.weight.double().cpu(), conv.bias.double().cpu(), stride, circular, up)
    res = torch.randn(ref.shape, device=dev)
    got = op.forward(x, up=up, res=res)
    assert_close(got, ref.detach() + res.double().cpu(), 2e-6, what='forward + residual')
    got_act = op.forward(x, up=up, act_in=1)
    ref_act = _ref_conv(F.silu(xd), conv.weight.double().cpu(), conv.bias.double().cpu(), stride, circular, up)
    assert_close(got_act, ref_act.detach(), 2e-6, what='loader activation')
    # input VJP: transposed convolution at the up-sampled resolution (the engine pools afterwards)
    cot = torch.randn(ref.shape, device=dev)
    xu = x
    for ax, u in enumerate(up):
        xu = xu.repeat_interleave(u, dim=2 + ax)
    xud = xu.double().cpu().requires_grad_(True)
    refu = _ref_conv(xud, conv.weight.double().cpu(), conv.bias.double().cpu(), stride, circular, (1, 1, 1))
    gref, = torch.autograd.grad((refu * cot.double().cpu()).sum(), xud)
    z = torch.randn(xu.shape, device=dev)
    radd = torch.randn(xu.shape, device=dev)
    gx = op.vjp(cot, tuple(xu.shape[2:]), res=radd)
    assert_close(gx, gref + radd.double().cpu(), 2e-6, what='VJP + residual')
    gz = op.vjp(cot, tuple(xu.shape[2:]), act=1, z=z)
    zd = z.double().cpu()
    sg = torch.sigmoid(zd)
    assert_close(gz, gref * (sg * (1 + zd * (1 - sg))), 2e-6, what="VJP x SiLU'(z)")
    if up != (1, 1, 1):
        from sda_amd.engine3d import _pool_sum
        gfull, = torch.autograd.grad((ref * cot.double().cpu()).sum(), xd)
        assert_close(_pool_sum(op.vjp(cot, tuple(xu.shape[2:])), up), gfull, 2e-6, what='pooled VJP of the up-sampled conv')


def test_unet3d_mid_size_against_oracle_and_guided_sampling(dev):
    """A wider three-level net (random init by seed) against the fp64 oracle: forward, input VJP, one guided evaluation
    (GaussianScore, autograd through A only) and a few PC steps."""
    from sda_amd.score import GaussianScore, ScoreUNet, VPSDE
    torch.manual_seed(5)
    net = ScoreUNet(3, context=1, embedding=16, hidden_channels=(12, 24, 40), hidden_blocks=(1, 2, 1), kernel_size=3,
                    activation=torch.nn.SiLU, spatial=3, padding_mode='circular').to(dev)
    cfg = O.UNetConfig(4, 3, 16, (12, 24, 40), (1, 2, 1), 3, 2, 'SiLU', 3, 'circular')
    sd = {k: v.double().cpu() for k, v in net.state_dict().items()}
    x = torch.randn(3, 3, 8, 12, 8, device=dev)
    c = torch.randn(1, 8, 12, 8, device=dev)
    t = torch.tensor(0.3, device=dev)
    xr = x.double().cpu().requires_grad_(True)
    ref = O.score_unet(sd, '', cfg, xr, t.double().cpu(), c.double().cpu())
    cot = torch.randn_like(x)
    gref, = torch.autograd.grad((ref * cot.double().cpu()).sum(), xr)
    xg = x.clone().requires_grad_(True)
    out = net(xg, t, c)
    assert_close(out, ref.detach(), 1e-5, what='forward')
    gx, = torch.autograd.grad((out * cot).sum(), xg)
    assert_close(gx, gref, 1e-5, what='input VJP')

    class WithContext(torch.nn.Module):                       # the forcing-channel pattern of LocalScoreUNet
        def __init__(self, inner, ctx):
            super().__init__()
            self.inner, self.ctx = inner, ctx

        def forward(self, x, t, c=None):
            return self.inner(x, t, self.ctx)

    A = lambda v: v[..., ::2, ::3, ::2]
    y = torch.randn(A(x).shape[1:], device=dev)
    gs = GaussianScore(y, A=A, std=0.3, sde=VPSDE(WithContext(net, c), shape=()), gamma=1e-2)
    got = gs(x, t)
    sched = O.Schedule()
    eps64 = lambda v, tt: O.score_unet(sd, '', cfg, v, tt, c.double().cpu())
    want = O.gaussian_score(eps64, sched, y.double().cpu(), A, 0.3, 1e-2, x.double().cpu(), t.double().cpu())
    assert_close(got, want, 1e-4, what='guided score')
    sde = VPSDE(gs, shape=(3, 8, 12, 8)).to(dev)
    torch.manual_seed(0)
    s = sde.sample((2,), steps=3, corrections=1, tau=0.5)
    assert s.shape == (2, 3, 8, 12, 8) and torch.isfinite(s).all()
    # the same loop with each step replayed from a hipGraph (PCSampler.capture)
    sde.use_graph = True
    torch.manual_seed(0)
    sg = sde.sample((2,), steps=3, corrections=1, tau=0.5)
    assert_close(sg, s, 1e-5, what='graph-replayed steps')


def test_mod_residual_block_3d_standalone(dev):
    """``ModResidualBlock`` called on its own (nn.py:18-28) with a per-sample modulation input."""
    net, g = _net('b', dev)
    blk = net.network.descent[1][0]
    torch.manual_seed(3)
    x = torch.randn(3, 20, 3, 3, 2, device=dev)
    y = torch.randn(3, 8, device=dev)
    got = blk(x, y)
    sd = {k: v.double().cpu() for k, v in blk.state_dict().items()}
    cfg = O.UNetConfig(20, 20, 8, (20,), (1,), (1, 3, 3), 1, 'ELU', 3, 'zeros')
    want = O._mod_block(sd, '', cfg, x.double().cpu(), y.double().cpu())
    assert_close(got, want, 1e-5)


def test_mcscorenet_over_a_3d_kernel(dev):
    """``MCScoreNet(..., spatial=3)`` (score.py:113-164 builds a ScoreUNet kernel for any ``spatial``): windows over a trajectory of
    volumes, one time per window, forward and input VJP against the oracle's composition."""
    from sda_amd.score import MCScoreNet
    torch.manual_seed(21)
    net = MCScoreNet(2, order=1, embedding=8, hidden_channels=(6, 12), hidden_blocks=(1, 1), kernel_size=3,
                     activation=torch.nn.SiLU, spatial=3, padding_mode='circular').to(dev)
    cfg = O.UNetConfig(6, 6, 8, (6, 12), (1, 1), 3, 2, 'SiLU', 3, 'circular')
    sd = {k: v.double().cpu() for k, v in net.state_dict().items()}
    x = torch.randn(2, 5, 2, 4, 4, 6, device=dev)
    t = torch.rand(2, 3, device=dev)
    kern = lambda a, b, _c=None: O.score_unet(sd, 'kernel.', cfg, a, b, None)
    xr = x.double().cpu().requires_grad_(True)
    ref = O.mc_score_net(kern, 1, xr, t.double().cpu())
    cot = torch.randn_like(x)
    gref, = torch.autograd.grad((ref * cot.double().cpu()).sum(), xr)
    xg = x.clone().requires_grad_(True)
    out = net(xg, t)
    assert_close(out, ref.detach(), 1e-5, what='forward')
    gx, = torch.autograd.grad((out * cot).sum(), xg)
    assert_close(gx, gref, 1e-5, what='input VJP')


def test_conv3d_batch_beyond_one_launch(dev, monkeypatch):
    """More images than one launch's grid takes (65 535 in the product; 3 here): the launcher walks the batch in slices."""
    from sda_amd.engine3d import _Conv3d
    torch.manual_seed(2)
    conv = torch.nn.Conv3d(3, 5, 3, padding=1).to(dev)
    x = torch.randn(8, 3, 4, 4, 4, device=dev)
    res = torch.randn(8, 5, 4, 4, 4, device=dev)
    op = _Conv3d(conv)
    whole = op.forward(x, res=res)
    monkeypatch.setattr(_Conv3d, 'MAX_IMAGES', 3)
    assert torch.equal(op.forward(x, res=res), whole)
    g = torch.randn_like(whole)
    z = torch.randn_like(x)
    sliced = op.vjp(g, (4, 4, 4), act=1, z=z)
    monkeypatch.setattr(_Conv3d, 'MAX_IMAGES', 65535)
    assert torch.equal(op.vjp(g, (4, 4, 4), act=1, z=z), sliced)
