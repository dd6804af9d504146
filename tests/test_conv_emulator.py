"""CPU: the conv_igemm tile algorithm replayed on the host (libsda_emu.so) against torch convolutions.

The emulator (bottom of sda_amd/csrc/conv_igemm.hip) shares the planner and every index helper with the gfx950
kernel and replays its staging / MFMA-fragment / epilogue maps lane by lane, so these tests pin the kernel's index
arithmetic (tiling, halo, circular wrap, stride, upsample, zero insertion, unfold view, context concat, LN/act
loader fusion, epilogue fusion) without a GPU.  The device kernel itself is tested in test_gpu_ops.py.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from oracle import sda_oracle as O
from sda_amd import build as sbuild
from sda_amd._lib import ConvDesc
from sda_amd.ops import CONV_CK, conv_out_size, make_conv_desc, pick_mt, round_up
from tests.util import assert_close


@pytest.fixture(scope='module')
def emu():
    lib = ctypes.CDLL(sbuild.build_emu())
    lib.sda_conv_igemm_emulate.restype = ctypes.c_int
    lib.sda_conv_igemm_emulate.argtypes = [ctypes.POINTER(ConvDesc)]
    lib.sda_pack_conv_weight_host.restype = None
    return lib


def pack(emu, w, transpose=False, cin_keep=None, mt=None):
    w = w.contiguous()
    cout, cin = w.shape[:2]
    ks = tuple(w.shape[2:])
    kh, kw = (1, ks[0]) if len(ks) == 1 else ks
    keep = cin if cin_keep is None else cin_keep
    k_real, m_real = (cout, keep) if transpose else (cin, cout)
    mt = mt or pick_mt(m_real)
    k_pad, m_pad = round_up(k_real, CONV_CK), round_up(m_real, 32 * mt)
    dst = torch.empty(kh * kw * k_pad * m_pad)
    emu.sda_pack_conv_weight_host(ctypes.c_void_p(w.data_ptr()), cout, cin, kh, kw, int(transpose), keep,
                                  ctypes.c_void_p(dst.data_ptr()), k_pad, m_pad)
    return dict(w=dst, k_pad=k_pad, m_pad=m_pad, mt=mt, kh=kh, kw=kw, m_real=m_real)


def run(emu, x, pk, ho, wo, *, n=None, cx=None, hs=None, ws=None, strides=None, **kw):
    """x: planar (n, c, h, w) unless explicit strides are given."""
    if strides is None:
        n, cx, hs, ws = x.shape
        strides = dict(x_sn_outer=x.stride(0), x_sc=x.stride(1), x_sy=x.stride(2), x_sx=x.stride(3))
    out = torch.full((n, pk['m_real'], ho, wo), float('nan'))
    keep = [x, out, pk['w']]
    ptr = {}
    for name in ('ctx', 'mod', 'ln_mean', 'ln_rstd', 'bias', 'dact_z', 'res'):
        t = kw.pop(name, None)
        if t is not None:
            t = t.contiguous()
            keep.append(t)
            ptr[name + '_ptr'] = t.data_ptr()
    d = make_conv_desc(x_ptr=x.data_ptr(), n=n, cx=cx, hs=hs, ws=ws, w_ptr=pk['w'].data_ptr(), cin_pad=pk['k_pad'],
                       cout_pad=pk['m_pad'], cout=pk['m_real'], kh=pk['kh'], kw=pk['kw'], out_ptr=out.data_ptr(),
                       ho=ho, wo=wo, mt=pk['mt'], **strides, **ptr, **kw)
    rc = emu.sda_conv_igemm_emulate(ctypes.byref(d))
    assert rc == 0, rc
    assert not torch.isnan(out).any()
    return out


def ref_conv(x, w, b, stride, circular):
    spatial = w.dim() - 2
    return O._conv(x, w, b, spatial, stride, 'circular' if circular else 'zeros')


@pytest.mark.parametrize('circular', [False, True])
@pytest.mark.parametrize('shape', [(3, 5, 8, 8, 7), (2, 11, 16, 16, 10), (1, 4, 5, 12, 3), (2, 9, 32, 32, 40),
                                   # tiny images: tiles of many images are cut down to what the loader covers
                                   (150, 4, 1, 1, 5), (70, 3, 3, 1, 4), (40, 2, 2, 5, 3)])
def test_conv2d_stride1(emu, circular, shape):
    n, cin, h, w_, cout = shape
    torch.manual_seed(0)
    x, w, b = torch.randn(n, cin, h, w_), torch.randn(cout, cin, 3, 3) * 0.2, torch.randn(cout)
    out = run(emu, x, pack(emu, w), h, w_, circular=circular, bias=b)
    assert_close(out, ref_conv(x, w, b, 1, circular), 2e-6)


@pytest.mark.parametrize('circular', [False, True])
def test_conv2d_stride2(emu, circular):
    torch.manual_seed(1)
    x, w, b = torch.randn(3, 6, 16, 16), torch.randn(12, 6, 3, 3) * 0.2, torch.randn(12)
    out = run(emu, x, pack(emu, w), 8, 8, circular=circular, bias=b, stride_h=2, stride_w=2)
    assert_close(out, ref_conv(x, w, b, 2, circular), 2e-6)


def test_conv2d_stride2_odd_zero_pad(emu):
    torch.manual_seed(2)
    x, w, b = torch.randn(2, 3, 9, 13), torch.randn(5, 3, 3, 3), torch.randn(5)
    ho, wo = conv_out_size(9, 3, 2), conv_out_size(13, 3, 2)
    out = run(emu, x, pack(emu, w), ho, wo, bias=b, stride_h=2, stride_w=2)
    assert_close(out, ref_conv(x, w, b, 2, False), 2e-6)


@pytest.mark.parametrize('circular', [False, True])
def test_upsample_fused(emu, circular):
    torch.manual_seed(3)
    x, w, b = torch.randn(2, 8, 4, 4), torch.randn(4, 8, 3, 3) * 0.2, torch.randn(4)
    up = x.repeat_interleave(2, -1).repeat_interleave(2, -2)
    out = run(emu, x, pack(emu, w), 8, 8, circular=circular, bias=b, up_h=2, up_w=2)
    assert_close(out, ref_conv(up, w, b, 1, circular), 2e-6)


@pytest.mark.parametrize('circular', [False, True])
@pytest.mark.parametrize('stride', [1, 2])
def test_backward_data_2d(emu, circular, stride):
    """transposed/flipped packing (+ zero insertion for stride 2) == autograd's input gradient."""
    torch.manual_seed(4)
    x = torch.randn(2, 5, 8, 8, requires_grad=True)
    w = torch.randn(7, 5, 3, 3) * 0.2
    y = ref_conv(x, w, None, stride, circular)
    g = torch.randn_like(y)
    gx_ref, = torch.autograd.grad(y, x, g)
    out = run(emu, g, pack(emu, w, transpose=True), 8, 8, circular=circular, zins_h=stride, zins_w=stride)
    assert_close(out, gx_ref, 2e-6)


def test_backward_data_drops_context_grads(emu):
    torch.manual_seed(5)
    x = torch.randn(2, 6, 8, 8, requires_grad=True)
    w = torch.randn(4, 6, 3, 3) * 0.2
    y = ref_conv(x, w, None, 1, True)
    g = torch.randn_like(y)
    gx_ref, = torch.autograd.grad(y, x, g)
    out = run(emu, g, pack(emu, w, transpose=True, cin_keep=5), 8, 8, circular=True)
    assert out.shape[1] == 5
    assert_close(out, gx_ref[:, :5], 2e-6)


@pytest.mark.parametrize('length', [16, 20, 65])
def test_conv1d(emu, length):
    torch.manual_seed(6)
    x, w, b = torch.randn(5, 3, length), torch.randn(8, 3, 3), torch.randn(8)
    x4 = x.unsqueeze(2)
    out = run(emu, x4, pack(emu, w), 1, length, bias=b)
    assert_close(out[:, :, 0], F.conv1d(x, w, b, padding=1), 2e-6)
    lo = conv_out_size(length, 3, 2)
    out = run(emu, x4, pack(emu, w), 1, lo, bias=b, stride_w=2)
    assert_close(out[:, :, 0], F.conv1d(x, w, b, padding=1, stride=2), 2e-6)
    # upsample on the length axis only
    out = run(emu, x4, pack(emu, w), 1, 2 * length, bias=b, up_w=2)
    assert_close(out[:, :, 0], F.conv1d(x.repeat_interleave(2, -1), w, b, padding=1), 2e-6)


def test_conv1d_backward_stride2_zero_pad(emu):
    torch.manual_seed(7)
    x = torch.randn(3, 4, 20, requires_grad=True)
    w = torch.randn(6, 4, 3)
    y = F.conv1d(x, w, None, stride=2, padding=1)
    g = torch.randn_like(y)
    gx_ref, = torch.autograd.grad(y, x, g)
    out = run(emu, g.unsqueeze(2), pack(emu, w, transpose=True), 1, 20, zins_w=2)
    assert_close(out[:, :, 0], gx_ref, 2e-6)


def test_transposed_layout_input(emu):
    """MCScoreWrapper's (B, L, C) tensor read as (B, C, L) through strides, no copy (score.py:104-110)."""
    torch.manual_seed(8)
    xt = torch.randn(4, 16, 3)                     # (B, L, C)
    w, b = torch.randn(8, 3, 3), torch.randn(8)
    strides = dict(x_sn_outer=xt.stride(0), x_sc=xt.stride(2), x_sy=0, x_sx=xt.stride(1))
    out = run(emu, xt, pack(emu, w), 1, 16, n=4, cx=3, hs=1, ws=16, strides=strides, bias=b)
    assert_close(out[:, :, 0], F.conv1d(xt.transpose(1, 2), w, b, padding=1), 2e-6)


def test_unfold_view_and_context_channel(emu):
    """head conv reading MCScoreNet windows straight out of (B, L, C, H, W) + broadcast forcing channel."""
    torch.manual_seed(9)
    B, L, C, H, W, k = 2, 7, 2, 8, 8, 2
    x = torch.randn(B, L, C, H, W)
    forcing = O.kolmogorov_forcing(8)
    wgt, b = torch.randn(6, (2 * k + 1) * C + 1, 3, 3) * 0.2, torch.randn(6)
    nw = L - 2 * k
    win = O.unfold(x, k)                                                    # (B, nw, 10, H, W)
    full = torch.cat((win, forcing.expand(B, nw, 1, H, W)), dim=2).reshape(B * nw, -1, H, W)
    ref = ref_conv(full, wgt, b, 1, True)
    strides = dict(x_sn_outer=x.stride(0), x_sn_inner=x.stride(1), n_inner=nw, x_sc=H * W, x_sy=W, x_sx=1)
    out = run(emu, x, pack(emu, wgt), H, W, n=B * nw, cx=(2 * k + 1) * C, hs=H, ws=W, strides=strides,
              circular=True, bias=b, ctx=forcing, cctx=1, ctx_sn=0)
    assert_close(out, ref, 2e-6)


@pytest.mark.parametrize('act', ['SiLU', 'GELU', 'ELU', 'ReLU', 'SELU'])
def test_loader_fusion_mod_layernorm_act(emu, act):
    from sda_amd._lib import ACT_IDS
    torch.manual_seed(10)
    n, c, h, w_ = 3, 12, 8, 8
    x, mod = torch.randn(n, c, h, w_) * 2 + 0.5, torch.randn(n, c)
    wgt, b = torch.randn(5, c, 3, 3) * 0.2, torch.randn(5)
    u = x + mod[:, :, None, None]
    var, mean = torch.var_mean(u, dim=1, unbiased=True, keepdim=True)
    rstd = 1 / torch.sqrt(var + 1e-5)
    hin = O.activation(act)(O.layer_norm(u, dim=1))
    ref = ref_conv(hin, wgt, b, 1, True)
    out = run(emu, x, pack(emu, wgt), h, w_, circular=True, bias=b, mod=mod, mod_sn=c,
              ln_mean=mean.reshape(n, -1), ln_rstd=rstd.reshape(n, -1), act_in=ACT_IDS[act])
    assert_close(out, ref, 5e-6)


def test_epilogue_dact_and_residual(emu):
    from sda_amd._lib import ACT_IDS
    torch.manual_seed(11)
    x, w = torch.randn(2, 6, 8, 8), torch.randn(9, 6, 3, 3) * 0.2
    z = torch.randn(2, 9, 8, 8, requires_grad=True)
    res = torch.randn(2, 9, 8, 8)
    dz, = torch.autograd.grad(F.silu(z).sum(), z)
    ref = ref_conv(x, w, None, 1, False) * dz + res
    out = run(emu, x, pack(emu, w), 8, 8, dact_z=z.detach(), act_d=ACT_IDS['SiLU'], res=res)
    assert_close(out, ref, 2e-6)


@pytest.mark.parametrize('cout,mt', [(96, 3), (40, 2), (130, 4)])
def test_multi_cout_tiles(emu, cout, mt):
    torch.manual_seed(12)
    x, w, b = torch.randn(1, 10, 8, 8), torch.randn(cout, 10, 3, 3) * 0.2, torch.randn(cout)
    out = run(emu, x, pack(emu, w, mt=mt), 8, 8, circular=True, bias=b)
    assert_close(out, ref_conv(x, w, b, 1, True), 2e-6)


def test_wide_image_and_5x5(emu):
    torch.manual_seed(13)
    x, w, b = torch.randn(1, 3, 2, 256), torch.randn(4, 3, 3, 3), torch.randn(4)
    assert_close(run(emu, x, pack(emu, w), 2, 256, circular=True, bias=b), ref_conv(x, w, b, 1, True), 2e-6)
    x, w = torch.randn(2, 3, 16, 16), torch.randn(4, 3, 5, 5) * 0.1
    assert_close(run(emu, x, pack(emu, w), 16, 16, circular=True), ref_conv(x, w, None, 1, True), 2e-6)


def test_winograd_zero_positions_of_upsampled_sources_and_pooled_outputs():
    """The algebra conv_wino4's zero-position kernels (ZP, csrc/conv_wino4.hip) rest on, in float64 on the host:
    F(2x2,3x3): Y = A^T [(G g G^T) . (B^T d B)] A.  (1) A 4 x 4 patch of a 2 x 2 nearest-upsampled image whose tile starts at an even
    pixel has equal rows 1, 2 and equal columns 1, 2: row / column 2 of B^T d B vanish EXACTLY (d2 - d1), so 7 of the 16 products are
    never needed.  (2) The sum of the tile's 2 x 2 outputs is v^T M v with v = (1, 2, 0, -1): the same 7 positions carry weight 0."""
    torch.manual_seed(0)
    Bt = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    At = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    dead = [(xi, nu) for xi in range(4) for nu in range(4) if xi == 2 or nu == 2]
    assert len(dead) == 7
    # (1) up-sampled source, float32 data: exact zeros
    src = torch.randn(3, 3, dtype=torch.float32)
    up = src.repeat_interleave(2, 0).repeat_interleave(2, 1)             # 6 x 6; a tile at even pixel (2, 2) reads rows / cols 1 .. 4
    d = up[1:5, 1:5]
    assert torch.equal(d[1], d[2]) and torch.equal(d[:, 1], d[:, 2])
    V = Bt.float() @ d @ Bt.float().T
    for xi, nu in dead:
        assert V[xi, nu].item() == 0.0
    # (2) pooled output of an arbitrary patch / filter
    dd, g = torch.randn(4, 4, dtype=torch.float64), torch.randn(3, 3, dtype=torch.float64)
    M = (G @ g @ G.T) * (Bt @ dd @ Bt.T)
    Y = At @ M @ At.T
    ref = F.conv2d(dd[None, None], g[None, None])[0, 0]                   # the direct 2 x 2 outputs
    assert torch.allclose(Y, ref, atol=1e-12)
    v = At.sum(0)
    assert v.tolist() == [1.0, 2.0, 0.0, -1.0]
    assert abs((v @ M @ v - Y.sum()).item()) < 1e-12
    Mz = M.clone()
    for xi, nu in dead:
        Mz[xi, nu] = float('nan')                                         # never read
    live = sum(v[xi] * v[nu] * Mz[xi, nu] for xi in (0, 1, 3) for nu in (0, 1, 3))
    assert abs((live - Y.sum()).item()) < 1e-12
