"""GPU: every C-ABI kernel against torch / the oracle on seeded inputs (fp32, tolerance 1e-4 scale-relative as
BASELINE.json's north_star states; typical measured error is ~1e-6)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sda_oracle as O
from tests.util import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def hip_conv(x, w, b, ho, wo, dev, transpose=False, cin_keep=None, **kw):
    from sda_amd import ops
    from sda_amd.engine import launch_conv, planar_source
    xd = x.to(dev).contiguous()
    pk = ops.PackedConv(w.to(dev), None if b is None else b.to(dev), transpose=transpose, cin_keep=cin_keep)
    out = torch.full((xd.shape[0], pk.m_real, ho, wo), float('nan'), device=dev)
    opts = {}
    for k, v in kw.items():
        opts[k] = v.to(dev).contiguous() if torch.is_tensor(v) else v
    if 'ln' in kw:
        opts['ln'] = tuple(t.to(dev).contiguous() for t in kw['ln'])
    launch_conv(pk, planar_source(xd), out, ho, wo, bias=pk.bias, **opts)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    return out.cpu()


def ref_conv(x, w, b, stride, circular):
    return O._conv(x, w, b, w.dim() - 2, stride, 'circular' if circular else 'zeros')


@pytest.mark.parametrize('circular', [False, True])
@pytest.mark.parametrize('shape', [(3, 5, 8, 8, 7), (2, 11, 64, 64, 96), (2, 96, 32, 32, 96), (1, 192, 16, 16, 384),
                                   (1, 4, 5, 12, 3), (2, 96, 64, 64, 10), (150, 4, 1, 1, 5), (70, 3, 3, 1, 4), (40, 2, 2, 5, 3)])
def test_conv2d_stride1(dev, circular, shape):
    n, cin, h, w_, cout = shape
    torch.manual_seed(0)
    x, w, b = torch.randn(n, cin, h, w_), torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5), torch.randn(cout)
    out = hip_conv(x, w, b, h, w_, dev, circular=circular)
    assert_close(out, ref_conv(x, w, b, 1, circular), TOL)


@pytest.mark.parametrize('circular', [False, True])
def test_conv2d_stride2(dev, circular):
    torch.manual_seed(1)
    x, w, b = torch.randn(3, 96, 64, 64), torch.randn(192, 96, 3, 3) * 0.03, torch.randn(192)
    out = hip_conv(x, w, b, 32, 32, dev, circular=circular, stride=(2, 2))
    assert_close(out, ref_conv(x, w, b, 2, circular), TOL)


@pytest.mark.parametrize('circular', [False, True])
def test_upsample_ln_fused(dev, circular):
    torch.manual_seed(3)
    x, w, b = torch.randn(2, 192, 16, 16) * 2 + 0.3, torch.randn(96, 192, 3, 3) * 0.03, torch.randn(96)
    skip = torch.randn(2, 96, 32, 32)
    hin = O.layer_norm(x, dim=1)
    up = hin.repeat_interleave(2, -1).repeat_interleave(2, -2)
    var, mean = torch.var_mean(x, dim=1, unbiased=True, keepdim=True)
    rstd = 1 / torch.sqrt(var + 1e-5)
    out = hip_conv(x, w, b, 32, 32, dev, circular=circular, up=(2, 2), ln=(mean.reshape(2, -1), rstd.reshape(2, -1)), res=skip)
    assert_close(out, ref_conv(up, w, b, 1, circular) + skip, TOL)


@pytest.mark.parametrize('circular', [False, True])
@pytest.mark.parametrize('stride', [1, 2])
def test_backward_data(dev, circular, stride):
    torch.manual_seed(4)
    x = torch.randn(2, 24, 16, 16, requires_grad=True)
    w = torch.randn(40, 24, 3, 3) * 0.1
    y = ref_conv(x, w, None, stride, circular)
    g = torch.randn_like(y)
    gx_ref, = torch.autograd.grad(y, x, g)
    out = hip_conv(g, w, None, 16, 16, dev, transpose=True, circular=circular, zins=(stride, stride))
    assert_close(out, gx_ref, TOL)


def test_conv1d_paths(dev):
    torch.manual_seed(6)
    for length in (16, 65, 128):
        x, w, b = torch.randn(5, 3, length), torch.randn(64, 3, 3), torch.randn(64)
        out = hip_conv(x.unsqueeze(2), w, b, 1, length, dev, circular=False)
        assert_close(out[:, :, 0], F.conv1d(x, w, b, padding=1), TOL)
    x = torch.randn(3, 4, 20, requires_grad=True)
    w = torch.randn(6, 4, 3)
    y = F.conv1d(x, w, None, stride=2, padding=1)
    g = torch.randn_like(y)
    gx_ref, = torch.autograd.grad(y, x, g)
    out = hip_conv(g.unsqueeze(2), w, None, 1, 20, dev, transpose=True, circular=False, zins=(1, 2))
    assert_close(out[:, :, 0], gx_ref, TOL)


@pytest.mark.parametrize('act', ['SiLU', 'GELU', 'ELU', 'ReLU', 'SELU'])
def test_loader_mod_ln_act_and_epilogue(dev, act):
    from sda_amd._lib import ACT_IDS
    torch.manual_seed(10)
    n, c, h, w_ = 3, 24, 16, 16
    x, mod = torch.randn(n, c, h, w_) * 2 + 0.5, torch.randn(n, c)
    wgt, b = torch.randn(20, c, 3, 3) * 0.1, torch.randn(20)
    u = x + mod[:, :, None, None]
    var, mean = torch.var_mean(u, dim=1, unbiased=True, keepdim=True)
    rstd = 1 / torch.sqrt(var + 1e-5)
    ref = ref_conv(O.activation(act)(O.layer_norm(u, dim=1)), wgt, b, 1, True)
    out = hip_conv(x, wgt, b, h, w_, dev, circular=True, mod=mod, mod_sn=c, ln=(mean.reshape(n, -1), rstd.reshape(n, -1)),
                   act_in=ACT_IDS[act])
    assert_close(out, ref, TOL)
    z = torch.randn(n, 20, h, w_, requires_grad=True)
    res = torch.randn(n, 20, h, w_)
    dz, = torch.autograd.grad(O.activation(act)(z).sum(), z)
    out = hip_conv(x, wgt, None, h, w_, dev, circular=True, dact_z=z.detach(), act_d=ACT_IDS[act], res=res)
    assert_close(out, ref_conv(x, wgt, None, 1, True) * dz + res, TOL)


@pytest.mark.parametrize('circular', [False, True])
@pytest.mark.parametrize('shape', [(2, 8, 6, 10, 5), (2, 192, 32, 32, 96), (1, 96, 16, 64, 192)])
@pytest.mark.parametrize('kern', [(1, 1), (1, 2), (2, 1), (2, 2)])
def test_conv_even_kernel_explicit_pad_strided_output(dev, circular, shape, kern):
    """The parity-class convolutions of the stride-2 VJP: 1- / 2-tap kernels with pad 0 (taps read in[o + t]), written
    into an interleaved view of a larger tensor together with a residual of the same layout."""
    from sda_amd import ops
    from sda_amd.engine import launch_conv, planar_source
    n, cin, h, w_, cout = shape
    kh, kw = kern
    torch.manual_seed(3)
    x, w = torch.randn(n, cin, h, w_), torch.randn(cout, cin, kh, kw) / (kh * kw * cin) ** 0.5
    xp = torch.nn.functional.pad(x, (0, kw - 1, 0, kh - 1), mode='circular' if circular else 'constant')
    ref = torch.nn.functional.conv2d(xp.double(), w.double()).float()
    big = torch.full((n, cout, 2 * h, 2 * w_), float('nan'), device=dev)
    res = torch.randn(n, cout, 2 * h, 2 * w_)
    resd = res.to(dev)
    pk = ops.PackedConv(w.to(dev), None)
    view = big[:, :, 1::2, 0::2]
    xd = x.to(dev)                                     # (held: planar_source keeps addresses, not tensors)
    launch_conv(pk, planar_source(xd), view, h, w_, circular=circular, pad=(0, 0), res=resd[:, :, 1::2, 0::2])
    torch.cuda.synchronize()
    assert_close(view.cpu(), ref + res[:, :, 1::2, 0::2], TOL)
    untouched = torch.ones(2 * h, 2 * w_, dtype=torch.bool)
    untouched[1::2, 0::2] = False
    assert torch.isnan(big.cpu()[:, :, untouched]).all()          # nothing outside the view was written


def test_ln_stats_apply_bwd(dev):
    from sda_amd import ops
    torch.manual_seed(11)
    for (n, c, h, w_), pool in (((3, 24, 8, 8), (1, 1)), ((2, 96, 16, 16), (2, 2)), ((4, 16, 1, 20), (1, 2)), ((2, 8, 1, 33), (1, 1)),
                              ((3, 8, 1, 6), (2, 2)),     # a 2-D net's deepest level may be one row high
                              # >= 16384 pixels: the thread-per-pixel kernels (below: wave-per-pixel)
                              ((5, 24, 64, 64), (1, 1)), ((3, 12, 64, 128), (2, 2)),
                              # register-resident statistics kernels: 1, 2 and 4 lanes per pixel (c <= 96, 192, 384)
                              ((5, 96, 64, 64), (1, 1)), ((5, 70, 64, 64), (1, 1)), ((3, 192, 64, 96), (1, 1)), ((3, 130, 64, 96), (2, 2)),
                              ((2, 384, 64, 65), (1, 1)), ((2, 300, 64, 65), (1, 1)), ((2, 96, 96, 96), (2, 2)), ((1, 384, 128, 130), (2, 2)), ((2, 6, 1, 9000), (1, 2)), ((2, 6, 1, 9000), (2, 2)),
                              # round 6: the exact-fit layouts of the 64 / 128 / 256-channel levels (c <= 64; 96 < c <= 128; 192 < c <= 256), full and ragged
                              ((5, 64, 64, 64), (1, 1)), ((5, 50, 64, 64), (1, 1)), ((3, 128, 64, 96), (1, 1)), ((3, 100, 64, 96), (1, 1)), ((2, 256, 64, 66), (1, 1)), ((2, 200, 64, 68), (1, 1)),
                              ((3, 128, 64, 96), (2, 2)), ((2, 64, 96, 96), (2, 2))):
        x = (torch.randn(n, c, h, w_) * 3 + 1).requires_grad_(True)
        mod = torch.randn(n, c)
        xd, md = x.detach().to(dev), mod.to(dev)
        mean = torch.empty(n * h * w_, device=dev); rstd = torch.empty_like(mean)
        ops.ln_stats(xd, md, c, 1e-5, True, mean, rstd)
        y = torch.empty_like(xd)
        ops.ln_apply(xd, md, c, mean, rstd, y)
        href = O.layer_norm(x + mod[:, :, None, None], dim=1)
        assert_close(y.cpu(), href, TOL, what='ln_apply')
        # backward (optionally through a nearest upsample)
        hup = href.repeat_interleave(pool[1], -1).repeat_interleave(pool[0], -2)
        g = torch.randn_like(hup)
        res = torch.randn(n, c, h, w_)
        gx_ref, = torch.autograd.grad(hup, x, g)
        gx = torch.empty_like(xd)
        ops.ln_bwd(g.to(dev), xd, h, w_, md, c, mean, rstd, True, pool, res.to(dev), gx)
        assert_close(gx.cpu(), gx_ref + res, TOL, what=f'ln_bwd pool={pool}')


def test_time_embed_and_projection(dev):
    from sda_amd import ops
    torch.manual_seed(12)
    sd = O.init_time_embedding(torch.Generator().manual_seed(3), 'e.', 64)
    t = torch.rand(7)
    ref = O.time_embedding(sd, 'e.', t)
    emb = ops.time_embed(t.to(dev), sd['e.freqs'].to(dev), sd['e.0.weight'].to(dev), sd['e.0.bias'].to(dev),
                         sd['e.2.weight'].to(dev), sd['e.2.bias'].to(dev))
    assert_close(emb.cpu(), ref, TOL)
    w, b = torch.randn(300, 64), torch.randn(300)
    y = ops.linear_small(emb, w.to(dev), b.to(dev))
    assert_close(y.cpu(), F.linear(ref, w, b), TOL)


def test_fold_and_adjoints(dev):
    from sda_amd import ops
    torch.manual_seed(13)
    for k, L in ((1, 5), (2, 9), (2, 5), (3, 7)):
        B, C, H, W = 2, 3, 4, 6
        nw, wl = L - 2 * k, 2 * k + 1
        s = torch.randn(B, nw, wl * C, H, W, requires_grad=True)
        ref = O.fold(s, k)
        out = torch.empty(B, L, C, H, W, device=dev)
        ops.fold(s.detach().to(dev).contiguous(), B, nw, k, C, H * W, out)
        assert torch.equal(out.cpu(), ref.detach())
        g = torch.randn_like(ref)
        gs_ref, = torch.autograd.grad(ref, s, g)
        gs = torch.empty(B, nw, wl * C, H, W, device=dev)
        ops.fold_adjoint(g.to(dev), B, nw, k, C, H * W, gs)
        assert torch.equal(gs.cpu(), gs_ref)
        x = torch.randn(B, L, C, H, W, requires_grad=True)
        u = O.unfold(x, k)
        gu = torch.randn_like(u)
        gx_ref, = torch.autograd.grad(u, x, gu)
        gx = torch.empty(B, L, C, H, W, device=dev)
        ops.unfold_adjoint(gu.to(dev).contiguous(), B, nw, k, C, H * W, wl * C, gx)
        assert_close(gx.cpu(), gx_ref, 1e-6)


def test_pc_updates_and_guidance_glue(dev):
    from sda_amd import ops
    torch.manual_seed(14)
    b, per = 3, 5000
    x, eps, z = torch.randn(b, per), torch.randn(b, per), torch.randn(b, per)
    xd = x.to(dev).clone()
    ops.pc_predict(xd, eps.to(dev), 1.25, -0.37)
    assert_close(xd.cpu(), 1.25 * x + (-0.37) * eps, 1e-6)
    partial = torch.empty(b * ops.SUMSQ_CHUNKS, device=dev)
    nch = ops.sumsq_partial(eps.to(dev), b, partial)           # (chunks per sample: one per 4096 elements, layout [b][nch])
    assert nch == ops.sumsq_chunks(per)
    assert_close(partial[:b * nch].reshape(b, nch).sum(1).cpu(), eps.square().sum(1), 1e-5)
    xd = x.to(dev).clone()
    ops.pc_correct(xd, eps.to(dev), z.to(dev), b, partial, 0.5, 0.8)
    delta = 0.5 / eps.square().mean(dim=1, keepdim=True)
    assert_close(xd.cpu(), x - (delta * eps + torch.sqrt(2 * delta) * z) * 0.8, 1e-5)
    xh = torch.empty(b, per, device=dev)
    ops.denoise(x.to(dev), eps.to(dev), 0.7, 0.6, xh)
    assert_close(xh.cpu(), (x - 0.6 * eps) / 0.7, 1e-6)
    mu_t, sg_t = torch.tensor(0.7, device=dev), torch.tensor(0.6, device=dev)
    xh2 = torch.empty(b, per, device=dev)
    ops.denoise(x.to(dev), eps.to(dev), mu_t, sg_t, xh2)
    assert torch.equal(xh, xh2)
    out = torch.empty(b, per, device=dev)
    ops.guided_combine(eps.to(dev), z.to(dev), x.to(dev), 0.7, 0.6, out)
    assert_close(out.cpu(), eps - 0.6 * (z / 0.7 - (0.6 / 0.7) * x), 1e-5)


def test_linear_and_row_layernorm(dev):
    from sda_amd import ops
    from sda_amd._lib import ACT_IDS
    torch.manual_seed(15)
    for rows, fin, fout in ((5, 23, 16), (300, 128, 128), (130, 256, 15), (64, 16, 200)):
        x, w, b = torch.randn(rows, fin), torch.randn(fout, fin) / fin ** 0.5, torch.randn(fout)
        y = ops.linear(x.to(dev), w.to(dev), b.to(dev))
        assert_close(y.cpu(), F.linear(x, w, b), TOL, what=f'linear {rows}x{fin}->{fout}')
        res = torch.randn(rows, fout)
        y = ops.linear(x.to(dev), w.to(dev), b.to(dev), act_in=ACT_IDS['SiLU'], res=res.to(dev))
        assert_close(y.cpu(), F.linear(F.silu(x), w, b) + res, TOL, what='linear act_in+res')
        g = torch.randn(rows, fout)
        z = torch.randn(rows, fin, requires_grad=True)
        dz, = torch.autograd.grad(F.silu(z).sum(), z)
        gx = ops.linear(g.to(dev), w.to(dev), None, trans_w=True, dact_z=z.detach().to(dev), act_d=ACT_IDS['SiLU'])
        assert_close(gx.cpu(), (g @ w) * dz, TOL, what='linear backward-data * act\'')
    for rows, f in ((7, 23), (100, 128), (33, 256), (5, 1000)):
        x = (torch.randn(rows, f) * 2 + 0.7).requires_grad_(True)
        xd = x.detach().to(dev)
        y = torch.empty_like(xd); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
        ops.row_ln(xd, 1e-5, True, y, mean, rstd)
        ref = O.layer_norm(x, dim=-1)
        assert_close(y.cpu(), ref, TOL, what='row_ln')
        g, res = torch.randn(rows, f), torch.randn(rows, f)
        gx_ref, = torch.autograd.grad(ref, x, g)
        gx = torch.empty_like(xd)
        ops.row_ln_bwd(g.to(dev), xd, mean, rstd, True, res.to(dev), gx)
        assert_close(gx.cpu(), gx_ref + res, TOL, what='row_ln_bwd')


def test_conv_randomised_sweep(dev):
    """A bounded sample of tests/fuzz/conv_fuzz.py (random layer shapes x random loader / epilogue fusions, every conv kernel
    family) -- the full sweep (2000 cases) is run by hand on the GPU box."""
    import importlib.util
    import os
    import random
    spec = importlib.util.spec_from_file_location(
        'conv_fuzz', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fuzz', 'conv_fuzz.py'))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    rng = random.Random(2024)
    failures = []
    for i in range(120):
        cfg, msg = fuzz.one_case(rng, dev, 50000 + i)
        if msg:
            failures.append((cfg, msg))
    assert not failures, failures[:3]


def test_vp_schedule_kernel(dev):
    """mu_sigma(t) for a device scalar == the schedule tables of every SDE flavour (oracle Schedule, pinned to the
    reference's own tables in tests/golden/schedule.npz), and the pair is handed to the kernels without a copy."""
    from sda_amd import ops
    from sda_amd.score import SubSubVPSDE, SubVPSDE, VPSDE
    for cls, kind in ((VPSDE, 'vp'), (SubVPSDE, 'subvp'), (SubSubVPSDE, 'subsubvp')):
        for alpha in ('cos', 'lin', 'exp'):
            sde = cls(torch.nn.Identity(), shape=(), alpha=alpha)
            sched = O.Schedule(alpha, kind=kind)
            for tv in (0.0, 0.013, 0.5, 0.987, 1.0):
                t = torch.tensor(tv)
                mu, sigma = sde.mu_sigma(t.to(dev))
                # (tolerance: mu = cos(k t)^2 near t = 1 is ~1e-3, where one ulp of the cosine is ~5e-6 of mu, and 1 - a^2 near t = 0 cancels -- for torch's own ops too: atol = a few fp32 ulps of 1)
                assert mu.shape == t.shape and ops._adjacent_pair(mu, sigma) is not None
                assert_close(mu.cpu(), sched.mu(t), 2e-5, atol=3e-6)
                assert_close(sigma.cpu(), sched.sigma(t), 2e-5, atol=3e-6)
                assert_close(mu.cpu(), sde.mu(t), 2e-5, atol=3e-6)          # == the torch-op path
                assert_close(sigma.cpu(), sde.sigma(t), 2e-5, atol=3e-6)
    # batched t and overridden schedules keep the generic path
    sde = VPSDE(torch.nn.Identity(), shape=())
    tb = torch.rand(4, device=dev)
    mu, sigma = sde.mu_sigma(tb)
    assert mu.shape == tb.shape and torch.equal(mu, sde.mu(tb)) and torch.equal(sigma, sde.sigma(tb))

    class Custom(VPSDE):
        def sigma(self, t):
            return t * 0 + 0.5
    assert float(Custom(torch.nn.Identity(), shape=()).mu_sigma(torch.tensor(0.3, device=dev))[1]) == 0.5


def test_small_operators_randomised_sweep(dev):
    """A bounded sample of tests/fuzz/ops_fuzz.py: observation operators (value, adjoint vs autograd, <A x, r> = <x, A^T r>),
    fold / unfold adjoints and the PC / guidance elementwise kernels on random shapes."""
    import importlib.util
    import os
    import random
    spec = importlib.util.spec_from_file_location(
        'ops_fuzz', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fuzz', 'ops_fuzz.py'))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    rng = random.Random(77)
    failures = []
    for i in range(90):
        cfg, msg = (fuzz.observe_case, fuzz.fold_case, fuzz.pc_case)[i % 3](rng, dev, 9000 + i)
        if msg:
            failures.append((cfg, msg))
    assert not failures, failures[:3]
    from sda_amd._lib import SdaHipError
    with pytest.raises(SdaHipError):                       # an undersized partial-sum buffer is refused, not overrun
        ops_ = __import__('sda_amd.ops', fromlist=['ops'])
        ops_.sumsq_partial(torch.randn(4, 10, device=dev), 4, torch.empty(4, 8, device=dev))


@pytest.mark.gpu
def test_gauss_cotangent_matches_torch(dev):
    """(y - A x)/var in one launch, against the reference's expression evaluated by torch (same operation order: bit exact),
    with y shared over the batch, per sample, and with (mu, sigma) as device scalars."""
    from sda_amd import ops
    torch.manual_seed(3)
    ax = torch.randn(5, 3, 7, 11, device=dev)
    std, gamma = 0.37, 1e-2
    for y in (torch.randn(5, 3, 7, 11, device=dev), torch.randn(1, 3, 7, 11, device=dev), torch.randn(3, 7, 11, device=dev)):
        for mu, sigma in ((0.83, 0.41), (torch.tensor(0.83, device=dev), torch.tensor(0.41, device=dev))):
            got = ops.gauss_cotangent(y, ax, std, gamma, mu, sigma)
            s_, m_ = torch.as_tensor(sigma, device=dev, dtype=torch.float32), torch.as_tensor(mu, device=dev, dtype=torch.float32)
            want = (y - ax) / (torch.tensor(std, device=dev) ** 2 + torch.tensor(gamma, device=dev) * (s_ / m_) ** 2)
            assert torch.equal(got, want)


@pytest.mark.gpu
def test_small_1d_conv_kernel_serves_the_lorenz_shapes(dev):
    """conv_small1d (one round trip per launch) against torch on the layer shapes of the Lorenz nets, every fusion on, both
    paddings, lengths that are not multiples of the tile; SDA_CONV_SMALL1D=0 is the staged kernel (compared too)."""
    import torch.nn.functional as F
    from sda_amd._lib import ACT_IDS
    torch.manual_seed(5)
    for (n, cin, cout, length, circular) in ((64, 40, 64, 128, False), (1, 3, 64, 64, True), (3, 64, 64, 77, True),
                                              (2, 64, 40, 130, False), (5, 17, 33, 16, True)):
        x = torch.randn(n, cin, 1, length)
        wgt = torch.randn(cout, cin, 3) / (3 * cin) ** 0.5
        b = torch.randn(cout)
        mod = torch.randn(1, cin)
        xin = x.double() + mod.double()[:, :, None, None]
        var, mean = torch.var_mean(xin, dim=1, unbiased=True, keepdim=True)
        rstd = 1 / torch.sqrt(var + 1e-5)
        xin = F.silu((xin - mean) * rstd)
        xp = F.pad(xin[:, :, 0], (1, 1), mode='circular' if circular else 'constant')
        ref = F.conv1d(xp, wgt.double(), b.double())[:, :, None]
        z = torch.randn(ref.shape)
        zz = z.double().requires_grad_(True)
        dz, = torch.autograd.grad(F.silu(zz).sum(), zz)
        res = torch.randn(ref.shape)
        ref = ref * dz + res.double()
        out = hip_conv(x, wgt[:, :, None], b, 1, length, dev, circular=circular, mod=mod,
                       ln=(mean.float().reshape(n, -1), rstd.float().reshape(n, -1)), act_in=ACT_IDS['SiLU'],
                       dact_z=z, act_d=ACT_IDS['SiLU'], res=res)
        assert_close(out.cpu(), ref.float(), 1e-4, what=f'small 1-D conv {(n, cin, cout, length, circular)}')


@pytest.mark.gpu
@pytest.mark.parametrize('circular', [True, False])
def test_winograd4_window_view_and_context_channel(dev, circular):
    """The Kolmogorov head convolution on the second-generation Winograd kernel: MCScoreNet windows read straight out of
    (B, L, C, H, W) (two-level image index), the forcing plane as a broadcast context channel, 11 -> 96 channels (a partial
    second K-stage); also a per-image context and a chunk offset."""
    from sda_amd import ops
    from sda_amd.engine import launch_conv
    torch.manual_seed(12)
    B, L, C, H, W, k = 2, 7, 2, 16, 32, 2
    x = torch.randn(B, L, C, H, W)
    wgt, b = torch.randn(96, (2 * k + 1) * C + 1, 3, 3) * 0.2, torch.randn(96)
    nw = L - 2 * k
    win = O.unfold(x, k)                                                    # (B, nw, 10, H, W)
    pk = ops.PackedConv(wgt.to(dev), b.to(dev))
    xd = x.to(dev)
    for ctx, ctx_sn in ((torch.randn(1, H, W), 0), (torch.randn(B * nw, 1, H, W), H * W)):
        cfull = ctx.expand(B * nw, 1, H, W) if ctx_sn == 0 else ctx
        full = torch.cat((win.reshape(B * nw, -1, H, W), cfull), dim=1)
        ref = ref_conv(full, wgt, b, 1, circular)
        for lo in (0, 2):
            n = B * nw - lo
            out = torch.full((n, 96, H, W), float('nan'), device=dev)
            src = dict(x_ptr=xd.data_ptr(), n=n, cx=(2 * k + 1) * C, hs=H, ws=W, x_sn_outer=xd.stride(0), x_sn_inner=xd.stride(1),
                       n_inner=nw, x_n_off=lo, x_sc=H * W, x_sy=W, x_sx=1)
            cd = ctx.to(dev).contiguous()
            cview = cd if ctx_sn == 0 else cd[lo:]
            desc = launch_conv(pk, src, out, H, W, circular=circular, bias=pk.bias, ctx=cview, cctx=1, ctx_sn=ctx_sn)
            torch.cuda.synchronize()
            assert ops.conv_path(desc) == 2, 'expected the second-generation Winograd kernel'
            assert_close(out.cpu(), ref[lo:], 1e-4, what=f'window view + context (ctx_sn={ctx_sn}, lo={lo})')


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(40, 96, 96, 32, 32, True), (9, 104, 96, 32, 64, False), (5, 192, 192, 32, 32, True),
                                   (3, 88, 96, 16, 32, True)])
def test_winograd4_epilogue_operand_through_the_helpers(dev, shape):
    """The residual add (sda/nn.py:28) and the multiply by act'(z) (backward of sda/nn.py:139) of conv_wino4 when the operand
    is staged by the helper waves (tiles of >= 12 K-stages): workgroups with several tiles (40 x 8 = 320 tiles on 256 CUs),
    a partial last stage (cin 104), two cout tiles per block (192), in-place residual (out aliases res) -- and a tile of
    11 stages (cin 88), which keeps the consumer-side loads."""
    from sda_amd import ops
    from sda_amd._lib import ACT_IDS
    from sda_amd.engine import launch_conv, planar_source
    n, cin, cout, h, w_, circular = shape
    torch.manual_seed(n + cin)
    x = torch.randn(n, cin, h, w_)
    wgt, b = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout)
    conv = ref_conv(x, wgt, b, 1, circular)
    res = torch.randn(n, cout, h, w_)
    z = torch.randn(n, cout, h, w_, requires_grad=True)
    dz, = torch.autograd.grad(F.silu(z).sum(), z)
    out = hip_conv(x, wgt, b, h, w_, dev, circular=circular, act_in=ACT_IDS['SiLU'], res=res)
    assert_close(out, ref_conv(F.silu(x), wgt, b, 1, circular) + res, TOL, what='silu + residual')
    out = hip_conv(x, wgt, b, h, w_, dev, circular=circular, dact_z=z.detach(), act_d=ACT_IDS['SiLU'])
    assert_close(out, conv * dz, TOL, what="x act'(z)")
    # in place: the output buffer is the residual
    pk = ops.PackedConv(wgt.to(dev), b.to(dev))
    xd, buf = x.to(dev), res.to(dev).clone()
    desc = launch_conv(pk, planar_source(xd), buf, h, w_, bias=pk.bias, circular=circular, res=buf)
    torch.cuda.synchronize()
    assert ops.conv_path(desc) == 2
    assert_close(buf.cpu(), conv + res, TOL, what='in-place residual')


@pytest.mark.gpu
@pytest.mark.parametrize('n,cin,cout,h,w_,circular', [(3, 96, 10, 64, 64, True), (2, 16, 1, 8, 32, False), (5, 32, 16, 16, 96, False),
                                                      (1, 48, 7, 40, 32, True), (130, 96, 10, 8, 32, True)])
def test_few_output_channel_kernel(dev, n, cin, cout, h, w_, circular):
    """conv_few.hip (3 x 3, <= 16 output channels: the U-Net tail sda/nn.py:166-176 and the head's backward-data form) against
    torch: both paddings, bias, residual, the backward-data packing with dropped input channels, a strided source -- and the
    shapes it must decline (ragged tile, loader fusion) still take the general kernels."""
    from sda_amd import ops
    from sda_amd._lib import ACT_IDS
    from sda_amd.engine import launch_conv, planar_source
    torch.manual_seed(n + cin + cout)
    x = torch.randn(n, cin, h, w_)
    wgt, b = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5, torch.randn(cout)
    ref = ref_conv(x, wgt, b, 1, circular)
    res = torch.randn_like(ref)
    pk = ops.PackedConv(wgt.to(dev), b.to(dev))
    xd = x.to(dev)
    out = torch.full((n, cout, h, w_), float('nan'), device=dev)
    desc = launch_conv(pk, planar_source(xd), out, h, w_, circular=circular, bias=pk.bias)
    torch.cuda.synchronize()
    assert ops.conv_path(desc) == 4, 'expected the few-output-channel kernel'
    assert_close(out.cpu(), ref, TOL, what='conv + bias')
    rd = res.to(dev)
    desc = launch_conv(pk, planar_source(xd), out, h, w_, circular=circular, bias=pk.bias, res=rd)
    assert ops.conv_path(desc) == 4
    assert_close(out.cpu(), ref + res, TOL, what='conv + bias + residual')
    # backward-data form of a (cout + 1 context channel) -> cin head, the context gradient dropped (cin_keep)
    wh = torch.randn(cin, cout + 1, 3, 3) / (9 * cin) ** 0.5
    pkb = ops.PackedConv(wh.to(dev), None, transpose=True, cin_keep=cout)
    xg = x.clone().requires_grad_(False)
    inp = torch.zeros(n, cout + 1, h, w_, requires_grad=True)
    yy = ref_conv(inp, wh, None, 1, circular)
    gref, = torch.autograd.grad(yy, inp, x)
    desc = launch_conv(pkb, planar_source(xd), out, h, w_, circular=circular)
    assert ops.conv_path(desc) == 4
    assert_close(out.cpu(), gref[:, :cout], TOL, what='backward-data, context channel dropped')
    # a strided source (channel-last memory)
    xcl = x.permute(0, 2, 3, 1).contiguous().to(dev)
    src = dict(x_ptr=xcl.data_ptr(), n=n, cx=cin, hs=h, ws=w_, x_sn_outer=xcl.stride(0), x_sc=1, x_sy=xcl.stride(1), x_sx=xcl.stride(2))
    desc = launch_conv(pk, src, out, h, w_, circular=circular, bias=pk.bias)
    assert ops.conv_path(desc) == 4
    assert_close(out.cpu(), ref, TOL, what='strided source')
    # declined shapes: a loader fusion, a ragged width
    desc = launch_conv(pk, planar_source(xd), out, h, w_, circular=circular, bias=pk.bias, act_in=ACT_IDS['SiLU'])
    assert ops.conv_path(desc) != 4
    assert_close(out.cpu(), ref_conv(F.silu(x), wgt, b, 1, circular), TOL, what='declined: activation in the loader')
    if w_ > 16:
        xr = xd[..., :w_ - 16].contiguous()
        outr = torch.empty(n, cout, h, w_ - 16, device=dev)
        desc = launch_conv(pk, planar_source(xr), outr, h, w_ - 16, circular=circular, bias=pk.bias)
        assert ops.conv_path(desc) != 4 or (w_ - 16) % 32 == 0
        assert_close(outr.cpu(), ref_conv(x[..., :w_ - 16], wgt, b, 1, circular), TOL, what='ragged width')


@pytest.mark.gpu
@pytest.mark.parametrize('n,cin,cout,h,w_,circular', [(2, 96, 192, 32, 64, True), (3, 192, 384, 16, 32, True), (1, 96, 96, 16, 32, False),
                                                      (5, 8, 96, 48, 32, True),
                                                      # the 64-cout tile (round 6): the heads 64 -> 128 and 128 -> 256 of the reference's default widths
                                                      (2, 64, 128, 32, 64, True), (3, 128, 256, 16, 32, False), (1, 320, 40, 16, 32, True), (2, 32, 64, 32, 64, True), (1, 160, 24, 16, 32, False)])
def test_stride2_vjp_one_launch_equals_four_classes_and_autograd(dev, monkeypatch, n, cin, cout, h, w_, circular):
    """conv_par4.hip: the backward-data of a stride-2 3 x 3 convolution (U-Net level heads, sda/nn.py:152-159) with all four
    output parity classes in one launch, against the four class launches and torch.autograd through the forward convolution;
    skip-gradient add fused.  (cin = the head's input channels = channels of the produced gradient; cout = its output channels.)"""
    import torch.nn as nn
    from sda_amd import ops
    from sda_amd.engine import _ConvCache, launch_conv, planar_source
    torch.manual_seed(n + cin + h)
    conv = nn.Conv2d(cin, cout, 3, stride=2, padding=1, padding_mode='circular' if circular else 'zeros')
    x = torch.randn(n, cin, h, w_, requires_grad=True)
    y = conv(x)
    g = torch.randn_like(y)
    gref, = torch.autograd.grad(y, x, g)
    skip = torch.randn(n, cin, h, w_)
    cc = _ConvCache(conv.to(dev))
    classes = cc.bwd_parity()
    gd, sd = g.to(dev), skip.to(dev)
    # four launches
    out4 = torch.full((n, cin, h, w_), float('nan'), device=dev)
    for py, px, pk, pad in classes:
        view = out4[:, :, py::2, px::2]
        launch_conv(pk, planar_source(gd), view, view.shape[2], view.shape[3], circular=circular, pad=pad, res=sd[:, :, py::2, px::2])
    assert_close(out4.cpu(), gref + skip, TOL, what='four class launches vs autograd')
    # one launch
    w4 = cc.bwd_parity4()
    eligible = cin % 32 == 0 and (h // 2) % 8 == 0 and (w_ // 2) % 16 == 0
    assert (w4 is not None) or not eligible
    out1 = torch.full((n, cin, h, w_), float('nan'), device=dev)
    v00 = out1[:, :, 0::2, 0::2]
    done = w4 is not None and launch_conv(classes[0][2], planar_source(gd), v00, v00.shape[2], v00.shape[3], circular=circular, pad=classes[0][3],
                                          res=sd[:, :, 0::2, 0::2], parity4_w=w4) is not None
    assert done == eligible, 'conv_par4 eligibility'
    if done:
        assert_close(out1.cpu(), gref + skip, TOL, what='one launch vs autograd')
        assert_close(out1.cpu(), out4.cpu(), 1e-5, what='one launch vs four launches')
        out1.fill_(float('nan'))
        launch_conv(classes[0][2], planar_source(gd), v00, v00.shape[2], v00.shape[3], circular=circular, pad=classes[0][3], parity4_w=w4)
        assert_close(out1.cpu(), gref, TOL, what='one launch, no skip operand')


@pytest.mark.gpu
def test_stride2_vjp_one_launch_random_shapes(dev):
    """A bounded random sweep of conv_par4 (every eligible shape class: 1-3 cout tiles of the produced gradient, K of 8-400 channels
    incl. partial 8-channel alignment, macro grids of 1-6 x 1-4 tiles, both paddings, with / without the skip operand) against
    torch.autograd through the forward stride-2 convolution."""
    import random
    import torch.nn as nn
    from sda_amd.engine import _ConvCache, launch_conv, planar_source
    rng = random.Random(7)
    for case in range(20):
        n = rng.choice([1, 2, 3])
        cin = rng.choice([96, 96, 192, 288, 64, 64, 128, 320, 32, 160])    # channels of the produced gradient (the head's input): 96-, 64- and 32-cout tiles
        cout = rng.choice([8, 24, 40, 96, 136, 192, 400])    # channels of the incoming gradient (the contraction)
        h, w_ = 16 * rng.randint(1, 4), 32 * rng.randint(1, 3)
        circular = rng.random() < 0.5
        torch.manual_seed(100 + case)
        conv = nn.Conv2d(cin, cout, 3, stride=2, padding=1, padding_mode='circular' if circular else 'zeros')
        x = torch.randn(n, cin, h, w_, requires_grad=True)
        y = conv(x)
        g = torch.randn_like(y)
        gref, = torch.autograd.grad(y, x, g)
        skip = torch.randn(n, cin, h, w_) if rng.random() < 0.6 else None
        cc = _ConvCache(conv.to(dev))
        classes, w4 = cc.bwd_parity(), cc.bwd_parity4()
        assert w4 is not None
        out = torch.full((n, cin, h, w_), float('nan'), device=dev)
        v00 = out[:, :, 0::2, 0::2]
        res = None if skip is None else skip.to(dev)[:, :, 0::2, 0::2]
        gd = g.to(dev)
        d = launch_conv(classes[0][2], planar_source(gd), v00, v00.shape[2], v00.shape[3], circular=circular, pad=classes[0][3],
                        res=res, parity4_w=w4)
        assert d is not None, (case, n, cin, cout, h, w_)
        want = gref if skip is None else gref + skip
        assert_close(out.cpu(), want, TOL, what=f'case {case}: n={n} {cout}->{cin} {h}x{w_} circular={circular} skip={skip is not None}')


@pytest.mark.gpu
@pytest.mark.parametrize('n,cin,cout,h,w_,circular', [(3, 96, 192, 32, 64, True), (2, 192, 384, 16, 32, False), (5, 96, 96, 8, 16, True),
                                                      (1, 40, 96, 24, 48, False)])
def test_pooled_output_is_the_upsample_vjp(dev, n, cin, cout, h, w_, circular):
    """sda_conv_desc.pool_h / pool_w (conv_wino4's pooled epilogue): the input VJP of Upsample(nearest, 2) -> conv3x3 (the U-Net
    tails, sda/nn.py:161-169) in one launch, against torch.autograd through the forward pair and against the two-step form (plain
    launch at the fine resolution, then 2 x 2 cell sums).  (cin = channels of the incoming gradient = the tail's outputs; cout =
    channels of the produced gradient = the tail's inputs; h, w_ = the FINE resolution.)"""
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
    launch_conv(cc.bwd(), planar_source(gd), fine, h, w_, circular=circular)
    two_step = 4 * F.avg_pool2d(fine, 2)
    assert_close(two_step.cpu(), gref, TOL, what='plain launch + cell sums vs autograd')
    pooled = torch.full((n, cout, h // 2, w_ // 2), float('nan'), device=dev)
    d = launch_conv(cc.bwd(), planar_source(gd), pooled, h, w_, circular=circular, pool=(2, 2))
    eligible = cout % 96 == 0 and h % 8 == 0 and w_ % 16 == 0
    assert (d is not None) == eligible, 'pooled-output eligibility'
    if d is not None:
        assert ops.conv_path(d) == 5
        assert_close(pooled.cpu(), gref, TOL, what='pooled launch vs autograd')
        assert_close(pooled.cpu(), two_step.cpu(), 2e-6, what='pooled launch vs plain launch + cell sums')


@pytest.mark.gpu
def test_pooled_output_refused_with_fusions_or_other_factors(dev):
    from sda_amd import ops
    from sda_amd.engine import launch_conv, planar_source
    x = torch.randn(2, 96, 16, 32, device=dev)
    pk = ops.PackedConv(torch.randn(96, 96, 3, 3, device=dev) * 0.05, None)
    out = torch.empty(2, 96, 8, 16, device=dev)
    assert launch_conv(pk, planar_source(x), out, 16, 32, circular=True, pool=(2, 2)) is not None
    assert launch_conv(pk, planar_source(x), out, 16, 32, circular=True, pool=(2, 2), res=torch.zeros_like(out)) is None
    assert launch_conv(pk, planar_source(x), out, 16, 32, circular=True, pool=(2, 2), act_in=1) is None
    assert launch_conv(pk, planar_source(x), torch.empty(2, 96, 16, 16, device=dev), 16, 32, circular=True, pool=(1, 2)) is None


@pytest.mark.gpu
@pytest.mark.parametrize('circular', [False, True])
def test_upsampled_tail_zero_position_kernel_is_bit_identical(dev, circular, tmp_path):
    """The zero-position form of conv_wino4 on a 2 x 2 up-sampled source skips products that are exact zeros: same bits as the full
    kernel (SDA_W4_ZP=0, read once per process -> a second process)."""
    import os
    import subprocess
    import sys
    script = f'''
import torch, sys
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from sda_amd import ops
from sda_amd.engine import launch_conv, planar_source
dev = torch.device('cuda:0')
torch.manual_seed(5)
x = (torch.randn(3, 192, 16, 24) * 2 + 0.3).to(dev)
pk = ops.PackedConv((torch.randn(96, 192, 3, 3) * 0.03).to(dev), torch.randn(96).to(dev))
skip = torch.randn(3, 96, 32, 48).to(dev)
var, mean = torch.var_mean(x, dim=1, unbiased=True, keepdim=True)
rstd = 1 / torch.sqrt(var + 1e-5)
out = torch.empty(3, 96, 32, 48, device=dev)
d = launch_conv(pk, planar_source(x), out, 32, 48, circular={circular}, up=(2, 2), ln=(mean.reshape(3, -1).contiguous(), rstd.reshape(3, -1).contiguous()),
                res=skip, bias=pk.bias)
torch.save((ops.conv_path(d), out.cpu()), sys.argv[1])
'''
    outs = []
    for zp in ('1', '0'):
        f = str(tmp_path / f'zp{zp}.pt')
        env = dict(os.environ, SDA_W4_ZP=zp)
        subprocess.run([sys.executable, '-c', script, f], check=True, env=env, timeout=600)
        outs.append(torch.load(f))
    assert outs[0][0] == 5 and outs[1][0] == 2, (outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.gpu
def test_zero_position_kernels_random_shapes(dev):
    """A bounded random sweep of conv_wino4's two zero-position launches (1-3 cout tiles, 12-50 K stages incl. partial 8-channel
    alignment, macro grids of 1-4 x 1-3 tiles, both paddings): the up-sampled LayerNorm + skip tail against torch, and the pooled VJP
    against torch.autograd through Upsample -> conv."""
    import random
    import torch.nn as nn
    from sda_amd import ops
    from sda_amd.engine import _ConvCache, launch_conv, planar_source
    rng = random.Random(7)
    for case in range(10):
        n = rng.choice([1, 2, 5])
        cin = rng.choice([96, 100, 192, 250, 384])
        cout = 96 * rng.choice([1, 1, 2, 3])
        h, w_ = 8 * rng.choice([1, 2, 3]), 16 * rng.choice([1, 2, 4])       # fine resolution
        circular = rng.random() < 0.5
        torch.manual_seed(case)
        # ---- forward tail: LN -> up -> conv (+ skip)
        x = torch.randn(n, cin, h // 2, w_ // 2) * 1.5 + 0.2
        wt, b = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5), torch.randn(cout)
        skip = torch.randn(n, cout, h, w_)
        var, mean = torch.var_mean(x, dim=1, unbiased=True, keepdim=True)
        rstd = 1 / torch.sqrt(var + 1e-5)
        up = O.layer_norm(x, dim=1).repeat_interleave(2, -1).repeat_interleave(2, -2)
        out = hip_conv(x, wt, b, h, w_, dev, circular=circular, up=(2, 2), ln=(mean.reshape(n, -1), rstd.reshape(n, -1)), res=skip)
        assert_close(out, ref_conv(up, wt, b, 1, circular) + skip, TOL, what=f'case {case}: up-sampled tail')
        # ---- its input VJP with the cell sums in the epilogue (produced channels = cout of the launch)
        conv = nn.Conv2d(cout, cin, 3, padding=1, padding_mode='circular' if circular else 'zeros')
        a = torch.randn(n, cout, h // 2, w_ // 2, requires_grad=True)
        y = conv(F.interpolate(a, scale_factor=2, mode='nearest'))
        g = torch.randn_like(y)
        gref, = torch.autograd.grad(y, a, g)
        pooled = torch.full((n, cout, h // 2, w_ // 2), float('nan'), device=dev)
        gd, cc = g.to(dev), _ConvCache(conv.to(dev))      # (held: planar_source keeps addresses, not tensors)
        d = launch_conv(cc.bwd(), planar_source(gd), pooled, h, w_, circular=circular, pool=(2, 2))
        assert d is not None and ops.conv_path(d) == 5, f'case {case}: pooled launch not served'
        assert_close(pooled.cpu(), gref, TOL, what=f'case {case}: pooled VJP')


@pytest.mark.parametrize('cin,hw', [(96, 32), (88, 16)])
def test_wino4_silu_derivative_strongly_negative_preactivation(dev, cin, hw):
    """the packed SiLU' of conv_wino4's epilogue (conv2^T x act'(z), backward of sda/nn.py:139) must stay
    finite and ~0 for z << 0 (e = exp(-z) overflows from z ~ -85), as sda_dact / torch's silu_backward do -- both the helper-fed
    (EPM = 1, >= 12 stages) and the consumer-side (EPM = 2, 11 stages) operand paths."""
    from sda_amd import ops
    from sda_amd._lib import ACT_IDS
    from sda_amd.engine import launch_conv, planar_source
    torch.manual_seed(cin)
    n, cout = 2, 96
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
    xd, zd = x.to(dev), z.to(dev)                        # (held: the descriptor carries raw pointers)
    desc = launch_conv(pk, planar_source(xd), out, hw, hw, circular=True, dact_z=zd, act_d=ACT_IDS['SiLU'])
    torch.cuda.synchronize()
    assert ops.conv_path(desc) == 2                      # conv_wino4
    assert torch.isfinite(out).all(), 'SiLU\'(z) overflowed for a strongly negative pre-activation'
    assert_close(out.cpu(), ref_conv(x, wgt, None, 1, True) * dz, TOL)


def test_clock_probe_reports_a_plausible_shader_clock(dev):
    """sda_clock_probe (measurement support for bench.py): the fp32 matrix-core stream's own clock, from the device's shader-clock
    and 100 MHz counters.  MI355X: 2.4 GHz peak, 2.1-2.4 sustained under this stream."""
    from sda_amd import ops
    r = ops.clock_probe(dev, ms=2.0)
    assert 1.2 < r['ghz_min'] <= r['ghz'] <= r['ghz_max'] < 2.6, r
    assert r['ghz_max'] - r['ghz_min'] < 0.5, r
