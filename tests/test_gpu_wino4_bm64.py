"""The 64-cout tile of conv_wino4 (template parameter MF = 2, round 6): Winograd F(2x2,3x3) for widths that are multiples of 64 but not of
96 -- the reference's DEFAULT Kolmogorov widths, `make_score(hidden_channels=(64, 128, 256))` (experiments/kolmogorov/utils.py:52), which
before ran every block convolution (sda/nn.py:131-142) and tail (sda/nn.py:161-169) on the direct kernel.  Every launch type against
float64 torch, the zero-position forms against the full kernel bit for bit, and the default-width net (forward, input VJP, guided score)
against the CPU oracle.  fp32, tolerance 1e-4 of the tensor scale (BASELINE.json north_star)."""
import importlib.util
import os
import random
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from oracle import sda_oracle as O
from tests.util import assert_close, oracle_eps_from_module

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def w4c(dev):
    spec = importlib.util.spec_from_file_location('wino4_check', os.path.join(ROOT, 'tools', 'wino4_check.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_conv(x, w, b, circular):
    xp = F.pad(x, (1, 1, 1, 1), mode='circular' if circular else 'constant')
    return F.conv2d(xp, w, b)


def test_structured_launch_types_on_the_64_cout_tile(w4c):
    """tools/wino4_check.py's structured cases with cout in {64, 128, 256, 320}: every loader configuration, the epilogue
    operand routes (none / consumer-side loads of one or two operands), partial last stages, 1-5 cout tiles,
    both paddings, the up-sampled tails, more tiles than workgroups -- each against float64, and on the kernel family expected."""
    cases = [c for c in w4c.structured() if c['cout'] % 96 != 0]                   # (incl. the 32-cout tile: cout 32 / 160)
    assert len(cases) >= 27
    for i, c in enumerate(cases):
        path, err, info = w4c.run_case(seed=4100 + i, **c)
        assert path == w4c.expect_path(c), (c, path)
        assert err <= TOL, (c, err, info)


def test_random_launches_on_the_64_cout_tile(w4c):
    rng = random.Random(64)
    worst = 0.0
    for i in range(40):
        c = dict(n=rng.choice([1, 2, 3, 5, 9]), cin=rng.choice([3, 8, 24, 40, 56, 64, 100, 128, 256]), cout=rng.choice([64, 64, 128, 256, 320, 32, 160]),
                 h=rng.choice([8, 16, 24, 32, 64]), w_=rng.choice([16, 32, 48, 64]), circular=rng.random() < 0.6, mod=rng.random() < 0.4,
                 ln=rng.random() < 0.4, silu=rng.random() < 0.4, up=rng.random() < 0.25, dact=rng.random() < 0.3, res=rng.random() < 0.4,
                 bias=rng.random() < 0.6)
        if c['cin'] * c['h'] * c['w_'] * c['n'] > 4e6:
            c['n'] = 1
        path, err, info = w4c.run_case(seed=6400 + i, **c)
        assert path == w4c.expect_path(c), (c, path)
        assert err <= TOL, (c, err, info)
        worst = max(worst, err)
    print(f'worst rel err {worst:.2e}')


@pytest.mark.parametrize('circular', [True, False])
def test_head_convolution_window_view_and_forcing_channel(dev, circular):
    """The default net's head: MCScoreNet windows of order 1 read straight out of (B, L, C, H, W), the forcing plane as a broadcast
    context channel, 7 -> 64 channels (one partial K-stage) -- experiments/kolmogorov/utils.py:29-59."""
    from sda_amd import ops
    from sda_amd.engine import launch_conv
    torch.manual_seed(21)
    B, L, C, H, W, k = 2, 6, 2, 16, 32, 1
    x = torch.randn(B, L, C, H, W)
    wgt, b = torch.randn(64, (2 * k + 1) * C + 1, 3, 3) * 0.2, torch.randn(64)
    nw = L - 2 * k
    win = O.unfold(x, k)
    ctx = torch.randn(1, H, W)
    full = torch.cat((win.reshape(B * nw, -1, H, W), ctx.expand(B * nw, 1, H, W)), dim=1)
    ref = ref_conv(full, wgt, b, circular)
    pk = ops.PackedConv(wgt.to(dev), b.to(dev))
    xd, cd = x.to(dev), ctx.to(dev).contiguous()
    for lo in (0, 3):
        n = B * nw - lo
        out = torch.full((n, 64, H, W), float('nan'), device=dev)
        src = dict(x_ptr=xd.data_ptr(), n=n, cx=(2 * k + 1) * C, hs=H, ws=W, x_sn_outer=xd.stride(0), x_sn_inner=xd.stride(1),
                   n_inner=nw, x_n_off=lo, x_sc=H * W, x_sy=W, x_sx=1)
        desc = launch_conv(pk, src, out, H, W, circular=circular, bias=pk.bias, ctx=cd, cctx=1, ctx_sn=0)
        torch.cuda.synchronize()
        assert ops.conv_path(desc) == 2
        assert_close(out.cpu(), ref[lo:], TOL, what=f'window view + forcing channel (lo={lo})')


@pytest.mark.parametrize('n,cin,cout,h,w_,circular', [(3, 64, 128, 32, 64, True), (2, 128, 256, 16, 32, False), (5, 64, 64, 8, 16, True),
                                                      (1, 40, 320, 24, 48, False), (3, 32, 32, 16, 32, True), (2, 64, 160, 16, 16, False)])
def test_pooled_output_is_the_upsample_vjp(dev, n, cin, cout, h, w_, circular):
    """The input VJP of Upsample(nearest, 2) -> conv3x3 (the tails, sda/nn.py:161-169) in one launch of the zero-position kernel at the
    64-cout tile (five 1-KiB slab pieces per helper): vs torch.autograd through the forward pair and vs the two-step form."""
    import torch.nn as nn
    from sda_amd import ops
    from sda_amd.engine import _ConvCache, launch_conv, planar_source
    torch.manual_seed(n + cin + h)
    conv = nn.Conv2d(cout, cin, 3, padding=1, padding_mode='circular' if circular else 'zeros')
    a = torch.randn(n, cout, h // 2, w_ // 2, requires_grad=True)
    y = conv(F.interpolate(a, scale_factor=2, mode='nearest'))
    g = torch.randn_like(y)
    gref, = torch.autograd.grad(y, a, g)
    cc = _ConvCache(conv.to(dev))
    gd = g.to(dev)
    fine = torch.full((n, cout, h, w_), float('nan'), device=dev)
    d0 = launch_conv(cc.bwd(), planar_source(gd), fine, h, w_, circular=circular)
    assert ops.conv_path(d0) == 2
    two_step = 4 * F.avg_pool2d(fine, 2)
    assert_close(two_step.cpu(), gref, TOL, what='plain launch + cell sums vs autograd')
    pooled = torch.full((n, cout, h // 2, w_ // 2), float('nan'), device=dev)
    d = launch_conv(cc.bwd(), planar_source(gd), pooled, h, w_, circular=circular, pool=(2, 2))
    assert d is not None and ops.conv_path(d) == 5
    assert_close(pooled.cpu(), gref, TOL, what='pooled launch vs autograd')
    assert_close(pooled.cpu(), two_step.cpu(), 2e-6, what='pooled launch vs plain launch + cell sums')


@pytest.mark.parametrize('circular', [False, True])
def test_upsampled_tail_zero_position_kernel_is_bit_identical(dev, circular, tmp_path):
    """The tail 128 -> 64 of the default net: the zero-position form skips products that are exact zeros -- same bits as the full
    64-cout kernel (SDA_W4_ZP=0 is read once per process -> a second process)."""
    script = f'''
import torch, sys
sys.path.insert(0, {ROOT!r})
from sda_amd import ops
from sda_amd.engine import launch_conv, planar_source
dev = torch.device('cuda:0')
torch.manual_seed(5)
x = (torch.randn(3, 128, 16, 24) * 2 + 0.3).to(dev)
pk = ops.PackedConv((torch.randn(64, 128, 3, 3) * 0.03).to(dev), torch.randn(64).to(dev))
skip = torch.randn(3, 64, 32, 48).to(dev)
var, mean = torch.var_mean(x, dim=1, unbiased=True, keepdim=True)
rstd = 1 / torch.sqrt(var + 1e-5)
out = torch.empty(3, 64, 32, 48, device=dev)
d = launch_conv(pk, planar_source(x), out, 32, 48, circular={circular}, up=(2, 2), ln=(mean.reshape(3, -1).contiguous(), rstd.reshape(3, -1).contiguous()),
                res=skip, bias=pk.bias)
torch.save((ops.conv_path(d), out.cpu()), sys.argv[1])
'''
    outs = []
    for zp in ('1', '0'):
        f = str(tmp_path / f'zp{zp}.pt')
        subprocess.run([sys.executable, '-c', script, f], check=True, env=dict(os.environ, SDA_W4_ZP=zp), timeout=600)
        outs.append(torch.load(f))
    assert outs[0][0] == 5 and outs[1][0] == 2, (outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('cin,hw', [(64, 32), (56, 16)])
def test_silu_derivative_strongly_negative_preactivation(dev, cin, hw):
    """conv2^T x act'(z) (backward of sda/nn.py:139) stays finite for z << 0 on both operand routes of the 64-cout tile: through the
    helpers (eight K-stages) and consumer-side loads (seven)."""
    from sda_amd import ops
    from sda_amd._lib import ACT_IDS
    from sda_amd.engine import launch_conv, planar_source
    torch.manual_seed(cin)
    n, cout = 2, 64
    x = torch.randn(n, cin, hw, hw)
    wgt = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    z = torch.randn(n, cout, hw, hw) * 3
    flat = z.view(-1)
    bad = torch.tensor([-85.0, -88.5, -89.5, -100.0, -1e4, -3e38, 85.0, 100.0, 1e4])
    flat[torch.randperm(flat.numel())[:bad.numel() * 40]] = bad.repeat(40)
    zz = z.clone().requires_grad_(True)
    dz, = torch.autograd.grad(F.silu(zz).sum(), zz)
    pk = ops.PackedConv(wgt.to(dev), None)
    out = torch.full((n, cout, hw, hw), float('nan'), device=dev)
    xd, zd = x.to(dev), z.to(dev)
    desc = launch_conv(pk, planar_source(xd), out, hw, hw, circular=True, dact_z=zd, act_d=ACT_IDS['SiLU'])
    torch.cuda.synchronize()
    assert ops.conv_path(desc) == 2
    assert torch.isfinite(out).all()
    assert_close(out.cpu(), ref_conv(x, wgt, None, True) * dz, TOL)


def _default_width_net(seed=3, size=64):
    from sda_amd.experiments.kolmogorov import make_score
    torch.manual_seed(seed)
    return make_score(size=size)                         # window 3, (64, 128, 256), (3, 3, 3): the reference's defaults, utils.py:49-57


def test_default_width_net_forward_and_vjp_vs_oracle(dev):
    """`make_score()` with the reference's default arguments on two 64 x 64 windows: eps against the fp32 oracle, J^T g against
    torch.autograd through the float64 oracle -- with the block convolutions and tails on conv_wino4's 64-cout tile (checked)."""
    from sda_amd import ops
    net = _default_width_net()
    eps_o = oracle_eps_from_module(net, 'mc2d')
    net.to(dev)
    torch.manual_seed(4)
    x = torch.randn(1, 4, 2, 64, 64)
    t = torch.tensor(0.35)
    g = torch.randn_like(x)
    xo = x.double().requires_grad_(True)
    eo = eps_o(xo, t.double(), torch.float64)
    ref, = torch.autograd.grad(eo, xo, g.double())
    prof = ops.ConvProfile()
    ops.conv_profile = prof
    try:
        xd = x.to(dev).requires_grad_(True)
        out = net(xd, t.to(dev))
        vjp, = torch.autograd.grad(out, xd, g.to(dev))
        torch.cuda.synchronize()
    finally:
        ops.conv_profile = None
    fams = prof.summary()['families']
    assert fams['wino4']['launches'] >= 2 * 36 + 1 and fams['wino4zp']['launches'] >= 4, {k: v['launches'] for k, v in fams.items()}
    with torch.no_grad():
        assert_close(out.detach().cpu(), eps_o(x, t), TOL, what='eps vs the fp32 oracle')
    assert_close(out.detach().cpu(), eo.detach(), TOL, what='eps vs the float64 oracle')
    assert_close(vjp.cpu(), ref, TOL, what='vjp')


def test_default_width_net_guided_score_and_pc_steps_vs_oracle(dev):
    """GaussianScore (sda/score.py:375-396) and two free-running predictor-corrector steps (score.py:250-261) of the default-width
    net at 32 x 32 against the oracle from the same draws."""
    from sda_amd.score import GaussianScore, VPSDE
    net = _default_width_net(seed=5, size=32)
    eps_o = oracle_eps_from_module(net, 'mc2d')
    A = lambda v: v[..., ::4, ::4]
    torch.manual_seed(6)
    x = torch.randn(2, 5, 2, 32, 32)
    y = torch.randn(A(x).shape)
    t = torch.tensor(0.6)
    sched = O.Schedule()
    ref = O.gaussian_score(lambda a, b: eps_o(a, b), sched, y, A, 0.3, 1e-2, x, t)
    net.to(dev)
    gs = GaussianScore(y, A=A, std=0.3, sde=VPSDE(net, shape=())).to(dev)
    out = gs(x.to(dev), t.to(dev)).cpu()
    assert_close(out, ref, TOL, what='guided score')
    sde = VPSDE(gs, shape=(5, 2, 32, 32)).to(dev)
    sde.initial_noise = x
    got = sde.sample((2,), steps=2, corrections=0)
    score = lambda a, b: O.gaussian_score(lambda p, q: eps_o(p, q), sched, y, A, 0.3, 1e-2, a, b)
    want = O.sample(score, sched, x, 4, steps=2, corrections=0, tau=1.0)
    assert_close(got.cpu(), want, TOL, what='two predictor steps')


def test_unet_default_width_net_32_64_128_forward_and_vjp_vs_oracle(dev):
    """`UNet`'s own default widths (sda/nn.py:99: hidden_channels (32, 64, 128)) in the Kolmogorov score net: the 32-cout tile (MF = 1)
    carries level 0, the 64-cout tile levels 1 and 2; eps against the fp32 oracle, J^T g against autograd through the float64 oracle."""
    from sda_amd import ops
    from sda_amd.experiments.kolmogorov import make_score
    torch.manual_seed(9)
    net = make_score(hidden_channels=(32, 64, 128), size=64)
    eps_o = oracle_eps_from_module(net, 'mc2d')
    net.to(dev)
    torch.manual_seed(10)
    x = torch.randn(1, 4, 2, 64, 64)
    t = torch.tensor(0.55)
    g = torch.randn_like(x)
    xo = x.double().requires_grad_(True)
    eo = eps_o(xo, t.double(), torch.float64)
    ref, = torch.autograd.grad(eo, xo, g.double())
    prof = ops.ConvProfile()
    ops.conv_profile = prof
    try:
        xd = x.to(dev).requires_grad_(True)
        out = net(xd, t.to(dev))
        vjp, = torch.autograd.grad(out, xd, g.to(dev))
        torch.cuda.synchronize()
    finally:
        ops.conv_profile = None
    fams = {k: v['launches'] for k, v in prof.summary()['families'].items()}
    assert fams.get('wino4', 0) >= 2 * 36 + 1 and fams.get('par4', 0) == 2, fams
    with torch.no_grad():
        assert_close(out.detach().cpu(), eps_o(x, t), TOL, what='eps vs the fp32 oracle')
    assert_close(vjp.cpu(), ref, TOL, what='vjp')
