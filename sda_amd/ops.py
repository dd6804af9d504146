"""Thin torch-tensor wrappers over the C ABI (include/sda_hip.h).

Every function launches a hand-written gfx950 kernel on the CURRENT torch stream.  There is no CPU path:
tensors must live on a HIP device (`_dev()` raises otherwise).
"""
import ctypes
import math
import os
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import ACT_IDS, ConvDesc

CONV_CK = 8          # K-stage depth of conv_igemm (SDA_CONV_CK)
WINOGRAD = os.environ.get('SDA_CONV_WINO', '1') != '0'      # Winograd F(2x2,3x3) for eligible 3x3 layers
WINOGRAD4 = os.environ.get('SDA_CONV_WINO4', '1') != '0'    # ... its one-wave-per-SIMD kernel where images are multiples of 16
#: OPT-IN: 'f16x2' = the block convolutions multiply on the f16 matrix cores with every fp32 operand split into two halves, fp32
#: accumulation (csrc/conv_h2.hip; same 3e-7 error against float64 as the fp32 Winograd kernel).  Default 'f32': fp32 MFMAs only.
MULTIPLY = os.environ.get('SDA_MULTIPLY', 'f32')
if MULTIPLY not in ('f32', 'f16x2'):
    raise ValueError(f"SDA_MULTIPLY={MULTIPLY!r} (expected 'f32' or 'f16x2')")


def set_multiply(mode: str) -> str:
    """Select how the block convolutions multiply: 'f32' (default: fp32 MFMAs) or 'f16x2' (opt-in, csrc/conv_h2.hip).  Returns the
    previous mode.  Takes effect at the next evaluation (packed weights are keyed on it); not inside a captured graph."""
    global MULTIPLY
    if mode not in ('f32', 'f16x2'):
        raise ValueError(f"multiply mode {mode!r} (expected 'f32' or 'f16x2')")
    prev, MULTIPLY = MULTIPLY, mode
    return prev


#: f16 x 2 route: the up-sampled tails on conv_h2's parity-class form (SDA_H2_UP=0: the zero-position Winograd kernel, A/B runs)
H2_UP = os.environ.get('SDA_H2_UP', '1') != '0'
#: ... and their pooled VJP on the parity-plane form (SDA_H2_POOL=0: the zero-position Winograd kernel)
H2_POOL = os.environ.get('SDA_H2_POOL', '1') != '0'
#: ... and the stride-2 heads with their VJP on the per-class tap lists (SDA_H2_S2=0: the fp32 direct / conv_par4 kernels)
H2_S2 = os.environ.get('SDA_H2_S2', '1') != '0'


def tensor_version(t) -> int:
    """``t._version`` for cache keys; inference-mode tensors have no version counter (reading it raises) and cannot be written
    in place either, so a constant stands in for them."""
    return 0 if t is None or t.is_inference() else t._version


def _dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.SdaHipError('sda_amd ops run on MI355X only: got a CPU tensor (there is no CPU fallback)')
        if t.dtype != torch.float32:
            raise _lib.SdaHipError(f'sda_amd ops are fp32: got {t.dtype}')


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def pick_mt(cout: int) -> int:
    """cout tile = 32*mt; choose the mt in 1..4 that wastes the fewest padded output channels."""
    best, best_pad = 1, None
    for mt in (3, 4, 2, 1):      # ties prefer 96-wide tiles: two double-buffered workgroups fit one CU's LDS
        pad = round_up(cout, 32 * mt)
        if best_pad is None or pad < best_pad:
            best, best_pad = mt, pad
    return best


def conv_out_size(size_v: int, k: int, stride: int) -> int:
    return (size_v + 2 * (k // 2) - k) // stride + 1


def make_conv_desc(*, x_ptr, n, cx, hs, ws, x_sc, x_sy, x_sx, x_sn_outer, x_sn_inner=0, n_inner=1, x_n_off=0,
                   w_ptr, cin_pad, cout_pad, cout, kh, kw, out_ptr, ho, wo, mt,
                   stride_h=1, stride_w=1, circular=False, up_h=1, up_w=1, zins_h=1, zins_w=1,
                   ctx_ptr=None, cctx=0, ctx_sn=0, mod_ptr=None, mod_sn=0, ln_mean_ptr=None, ln_rstd_ptr=None,
                   act_in=0, bias_ptr=None, dact_z_ptr=None, act_d=0, res_ptr=None, w_wino_ptr=None,
                   w_wino4_ptr=None, pad=None, out_strides=(0, 0, 0, 0), pool=(1, 1), w_wino4_zp_ptr=None) -> ConvDesc:
    d = ConvDesc()
    d.x = x_ptr
    d.x_sn_outer, d.x_sn_inner, d.n_inner, d.x_n_off = x_sn_outer, x_sn_inner, n_inner, x_n_off
    d.x_sc, d.x_sy, d.x_sx = x_sc, x_sy, x_sx
    d.cx = cx
    d.ctx, d.ctx_sn, d.cctx = ctx_ptr, ctx_sn, cctx
    d.n, d.hs, d.ws = n, hs, ws
    d.up_h, d.up_w, d.zins_h, d.zins_w = up_h, up_w, zins_h, zins_w
    d.mod, d.mod_sn = mod_ptr, mod_sn
    d.ln_mean, d.ln_rstd = ln_mean_ptr, ln_rstd_ptr
    d.act_in = act_in
    d.kh, d.kw, d.stride_h, d.stride_w, d.circular = kh, kw, stride_h, stride_w, int(bool(circular))
    d.w, d.cin_pad, d.cout_pad = w_ptr, cin_pad, cout_pad
    d.bias = bias_ptr
    d.out, d.cout, d.ho, d.wo = out_ptr, cout, ho, wo
    d.dact_z, d.act_d = dact_z_ptr, act_d
    d.res = res_ptr
    d.mt = mt
    d.w_wino = w_wino_ptr
    d.w_wino4 = w_wino4_ptr
    d.explicit_pad = 0 if pad is None else 1
    d.pad_h, d.pad_w = (0, 0) if pad is None else pad
    d.out_sn, d.out_sc, d.out_sy, d.out_sx = out_strides
    d.pool_h, d.pool_w = pool
    d.w_wino4_zp = w_wino4_zp_ptr
    return d


class ConvProfile:
    """Live timing of the kernels for bench.py's roofline leg: when installed as ``ops.conv_profile`` every conv_igemm
    launch is bracketed by HIP events on the launch stream and its ALGORITHMIC flops are recorded (2 * n * out-pixels *
    cout * cin * taps over the real channels; zero-inserted taps are not counted), by kernel family; the HBM-bound LayerNorm
    kernels are bracketed the same way with their algorithmic bytes (one read / write per operand)."""

    def __init__(self):
        self.records = []          # (start_event, stop_event, flops, family)
        self.family_bytes = {}     # family -> [algorithmic bytes read, written] over all its launches
        self.mem_records = []      # (start_event, stop_event, bytes, kernel)

    def flops(self, d: ConvDesc) -> float:
        pixels = d.ho * d.wo / (d.zins_h * d.zins_w)
        return 2.0 * d.n * pixels * d.cout * (d.cx + d.cctx) * d.kh * d.kw

    def alg_bytes(self, d: ConvDesc):
        """ALGORITHMIC bytes of a convolution launch: every operand read once (source, weights, LayerNorm statistics, epilogue
        operands), the output written once -- what the PMC traffic of profiles/*_traffic.json is compared against."""
        cin = d.cx + d.cctx
        rd = 4.0 * d.n * d.cx * d.hs * d.ws + 4.0 * d.cctx * d.hs * d.ws * (d.n if d.ctx_sn else 1)
        rd += 4.0 * cin * d.cout * (16 if (d.w_wino4 or d.w_wino) and conv_path(d) in (1, 2, 5) else d.kh * d.kw)
        if d.ln_mean:
            rd += 8.0 * d.n * d.hs * d.ws
        out = 4.0 * d.n * d.cout * d.ho * d.wo / (max(1, d.pool_h) * max(1, d.pool_w))
        rd += out * ((1 if d.res else 0) + (1 if d.dact_z else 0))
        return rd, out

    def bracket_mem(self, kernel: str, nbytes: float, launch):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        self.mem_records.append((e0, e1, nbytes, kernel))

    def mem_summary(self):
        torch.cuda.synchronize()
        out = {}
        for a, b, nb, k in self.mem_records:
            r = out.setdefault(k, dict(launches=0, ms=0.0, bytes=0.0))
            r['launches'] += 1
            r['ms'] += a.elapsed_time(b)
            r['bytes'] += nb
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = dict(launches=len(self.records), total_ms=0.0, total_flops=0.0, families={})
        for a, b, f, fam in self.records:
            ms = a.elapsed_time(b)
            out['total_ms'] += ms
            out['total_flops'] += f
            r = out['families'].setdefault(fam, dict(launches=0, ms=0.0, flops=0.0))
            r['launches'] += 1
            r['ms'] += ms
            r['flops'] += f
        return out


conv_profile: Optional[ConvProfile] = None


def conv_igemm(desc: ConvDesc):
    lib = _lib.load()
    prof = conv_profile
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.sda_conv_igemm(ctypes.byref(desc), _stream()), 'sda_conv_igemm')
        e1.record()
        fam = CONV_FAMILIES[max(0, conv_path(desc))]
        # (a zero-insertion launch multiplies a quarter of what its fine-resolution output size says: 9 taps per SOURCE pixel)
        prof.records.append((e0, e1, prof.flops(desc) / float(max(1, desc.zins_h) * max(1, desc.zins_w)), fam))
        rd, wr = prof.alg_bytes(desc)
        b = prof.family_bytes.setdefault(fam, [0.0, 0.0])
        b[0] += rd
        b[1] += wr
        return
    _lib.check(lib.sda_conv_igemm(ctypes.byref(desc), _stream()), 'sda_conv_igemm')


def absmax(x: Tensor, out: Tensor) -> Tensor:
    """out[0] = max |x| (device scalar; one streaming read of x): the input scale of an f16x2 launch whose producer did not report it."""
    _dev(x, out)
    if not x.is_contiguous():
        raise _lib.SdaHipError('absmax reads a contiguous tensor')
    _lib.check(_lib.load().sda_absmax(x.data_ptr(), x.numel(), out.data_ptr(), _stream()), 'sda_absmax')
    return out


def conv_h2(desc: ConvDesc, pk: 'PackedConv', x_amax, out_amax: Optional[Tensor], packing=None) -> bool:
    """Run the launch on the f16 x 2 kernel (csrc/conv_h2.hip) when `pk` carries that packing and the kernel serves the shape.
    x_amax: device scalar (Tensor) or a host bound (float) on the magnitude of what the loader feeds the multiply.  False: not
    served -- the caller runs the fp32 kernels."""
    if getattr(pk, 'h2', None) is None or x_amax is None:
        return False
    lib = _lib.load()
    if packing is None:
        desc.w_h2, desc.w_h2_scale = pk.h2.data_ptr(), pk.h2_scale
    else:                                                 # (buffer, scale): e.g. PackedConv.h2_up() for an up-sampled source
        desc.w_h2, desc.w_h2_scale = packing[0].data_ptr(), packing[1]
    if torch.is_tensor(x_amax):
        desc.x_amax, desc.x_amax_static = x_amax.data_ptr(), 0.0
    else:
        desc.x_amax, desc.x_amax_static = None, float(x_amax)
    desc.out_amax = None if out_amax is None else out_amax.data_ptr()
    if not lib.sda_conv_h2_supported(ctypes.byref(desc)):
        desc.w_h2 = None
        return False
    prof = conv_profile
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if out_amax is not None:
        out_amax.zero_()
    _lib.check(lib.sda_conv_h2(ctypes.byref(desc), _stream()), 'sda_conv_h2')
    if prof is not None:
        e1.record()
        # (the up-sampled / pooled forms issue 4 of the 9 taps; the stride-2 forms all 9, at a quarter of the fine-resolution pixels)
        fam = 'h2up' if (desc.up_h == 2 or desc.pool_h == 2) else 'h2s2' if (desc.zins_h == 2 or desc.stride_h == 2) else 'h2'
        # (a zero-insertion launch multiplies a quarter of what its fine-resolution output size says: 9 taps per SOURCE pixel)
        prof.records.append((e0, e1, prof.flops(desc) / float(max(1, desc.zins_h) * max(1, desc.zins_w)), fam))
        rd, wr = prof.alg_bytes(desc)
        b = prof.family_bytes.setdefault(fam, [0.0, 0.0])
        b[0] += rd
        b[1] += wr
    return True


PARITY4 = os.environ.get('SDA_CONV_PAR4', '1') != '0'
WINO4_BM64 = os.environ.get('SDA_W4_BM64', '1') != '0'      # 0: widths that are multiples of 32 but not of 96 stay on the direct kernel (A/B runs)


def conv_parity4(desc: ConvDesc) -> bool:
    """The four parity classes of a stride-2 3 x 3 convolution's backward-data in one launch (csrc/conv_par4.hip).  False: the
    shape is outside the kernel's range (the caller runs the four class launches)."""
    lib = _lib.load()
    prof = conv_profile
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib.sda_conv_parity4(ctypes.byref(desc), _stream())
    if rc == -2:                                         # SDA_E_UNSUPPORTED
        return False
    _lib.check(rc, 'sda_conv_parity4')
    if prof is not None:
        e1.record()
        prof.records.append((e0, e1, 2.0 * desc.n * desc.ho * desc.wo * desc.cout * desc.cx * 9, 'par4'))
        b = prof.family_bytes.setdefault('par4', [0.0, 0.0])
        b[0] += 4.0 * desc.n * desc.cx * desc.hs * desc.ws + 4.0 * 9 * desc.cx * desc.cout + (16.0 * desc.n * desc.cout * desc.ho * desc.wo if desc.res else 0.0)
        b[1] += 16.0 * desc.n * desc.cout * desc.ho * desc.wo
    return True


POOLED = os.environ.get('SDA_CONV_POOLED', '1') != '0'


def conv_pooled(desc: ConvDesc) -> bool:
    """A convolution whose output is summed over pool_h x pool_w cells (sda_conv_desc.pool_h / pool_w).  False: no kernel serves
    the pooled form of this launch (the caller runs the plain launch and pools in its reader)."""
    lib = _lib.load()
    if lib.sda_conv_igemm_path(ctypes.byref(desc)) < 0:          # SDA_E_UNSUPPORTED: planning only, nothing launched
        return False
    conv_igemm(desc)
    return True


CONV_FAMILIES = ('direct', 'wino', 'wino4', 'small1d', 'few', 'wino4zp')     # indexed by sda_conv_igemm_path


def conv_path(desc: ConvDesc) -> int:
    """Kernel family that would serve the launch: 2 one-wave-per-SIMD Winograd, 5 its zero-position form (2 x 2 up-sampled source or
    pooled output), 1 Winograd, 3 small 1-D kernel, 4 few-output-channel kernel, 0 direct implicit GEMM."""
    return _lib.load().sda_conv_igemm_path(ctypes.byref(desc))


def pack_conv_weight(w: Tensor, cout: int, cin: int, kh: int, kw: int, transpose: bool, keep: int, dst: Tensor,
                     k_pad: int, m_pad: int):
    _lib.check(_lib.load().sda_pack_conv_weight(w.data_ptr(), cout, cin, kh, kw, int(transpose), keep, dst.data_ptr(),
                                                k_pad, m_pad, _stream()), 'sda_pack_conv_weight')


class PackedConv:
    """A conv layer's weights repacked for sda_conv_igemm (forward or backward-data form)."""

    def __init__(self, weight: Tensor, bias: Optional[Tensor], transpose: bool = False, cin_keep: Optional[int] = None):
        _dev(weight, bias)
        w = weight.detach().contiguous()
        cout, cin = w.shape[0], w.shape[1]
        ks = tuple(w.shape[2:])
        self.kh, self.kw = (1, ks[0]) if len(ks) == 1 else ks
        self._transpose = bool(transpose)
        self._w_ref = w if (MULTIPLY == 'f16x2' and len(ks) == 2) else None     # (h2_up() sums taps of the original layout on first use)
        self._h2_up = None
        self._h2_pool = None
        self._h2_zins = None
        self._h2_s2 = None
        if transpose:
            keep = cin if cin_keep is None else cin_keep
            self.k_real, self.m_real = cout, keep          # contraction over forward cout, produces forward cin
        else:
            keep = cin
            self.k_real, self.m_real = cin, cout
        self.mt = pick_mt(self.m_real)
        self.k_pad = round_up(self.k_real, CONV_CK)
        self.m_pad = round_up(self.m_real, 32 * self.mt)
        self.packed = torch.empty(self.kh * self.kw * self.k_pad * self.m_pad, device=w.device, dtype=torch.float32)
        pack_conv_weight(w, cout, cin, self.kh, self.kw, transpose, keep, self.packed, self.k_pad, self.m_pad)
        self.bias = None if (bias is None or transpose) else bias.detach().contiguous()
        # Winograd F(2x2,3x3) form for 3x3 layers whose output channels tile by 96 (the U-Net block / tail convolutions)
        self.wino = None
        if WINOGRAD and (self.kh, self.kw) == (3, 3) and self.m_real % 96 == 0 and self.mt == 3:
            self.wino = torch.empty(16 * self.k_pad * self.m_pad, device=w.device, dtype=torch.float32)
            _lib.check(_lib.load().sda_pack_conv_weight_wino(w.data_ptr(), cout, cin, int(transpose), keep,
                                                             self.wino.data_ptr(), self.k_pad, self.m_pad, _stream()),
                       'sda_pack_conv_weight_wino')
        # ... and the packing of the second-generation Winograd kernel (conv_wino4.hip: U fragments in MFMA lane order).  Its cout tile is
        # 96 where the width allows (the reference's training widths (96, 192, 384)), 64 for the other multiples of 64 -- the reference's
        # DEFAULT widths (64, 128, 256), experiments/kolmogorov/utils.py:52 -- and 32 for the remaining multiples of 32 (UNet's own
        # default (32, 64, 128), sda/nn.py:99) (round 6; the first-generation kernel stays 96-only)
        self.wino4 = None
        if WINOGRAD4 and (self.kh, self.kw) == (3, 3) and self.m_pad == self.m_real and (self.wino is not None or (WINO4_BM64 and self.m_real % 32 == 0)):
            self.wino4 = torch.empty(16 * self.k_pad * self.m_pad, device=w.device, dtype=torch.float32)
            _lib.check(_lib.load().sda_pack_conv_weight_wino4(w.data_ptr(), cout, cin, int(transpose), keep,
                                                              self.wino4.data_ptr(), self.k_pad, self.m_pad, _stream()),
                       'sda_pack_conv_weight_wino4')
        # ... and its zero-position packing (sda_pack_conv_weight_wino4_zp: the 9 live Winograd positions of a 2 x 2 up-sampled / pooled
        # launch, 28 KiB per K stage and cout tile instead of 36) -- made here, not on first use: a launch must not allocate (its operands
        # may be views of freed temporaries the allocator would hand out again; a step may be under graph capture)
        self._wino4_zp = None
        if self.wino4 is not None:
            lib = _lib.load()
            self._wino4_zp = torch.empty(int(lib.sda_wino4_zp_floats(self.k_pad, self.m_pad)), device=w.device, dtype=torch.float32)
            _lib.check(lib.sda_pack_conv_weight_wino4_zp(self.wino4.data_ptr(), self.k_pad, self.m_pad, self._wino4_zp.data_ptr(), _stream()),
                       'sda_pack_conv_weight_wino4_zp')

        # ... and, OPT-IN (ops.MULTIPLY == 'f16x2'), the two-halves packing of csrc/conv_h2.hip with its scale (max |w| is read back
        # once, here -- packing happens in warm-up, never under graph capture) and the device scalars its launches report through
        self.h2 = None
        if MULTIPLY == 'f16x2' and (self.kh, self.kw) == (3, 3) and (not transpose or keep == cin) and w.numel() > 0:
            lib = _lib.load()
            nbytes = int(lib.sda_conv_h2_packed_bytes(cout, cin, int(transpose)))
            if nbytes > 0:
                w_amax = float(w.abs().max())
                if w_amax > 0.0 and math.isfinite(w_amax):
                    self.h2 = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
                    _lib.check(lib.sda_pack_conv_weight_h2(w.data_ptr(), cout, cin, int(transpose), w_amax, self.h2.data_ptr(), _stream()),
                               'sda_pack_conv_weight_h2')
                    self.h2_scale = float(lib.sda_conv_h2_scale(w_amax))
                    self.out_amax = torch.zeros(1, device=w.device, dtype=torch.float32)     # max |out| of this layer's last launch
                    self.in_amax = torch.zeros(1, device=w.device, dtype=torch.float32)      # scratch for an absmax pass over its input

    def _h2_wsum(self) -> Tensor:
        """[4 classes (2 py + px)][cout][cin][4 taps (2 a + b)]: the 3 x 3 taps that read ONE source pixel of a 2 x 2 nearest-up-sampled image,
        summed (fp32) -- output pixel (2 i + py, 2 j + px) sees source rows i - 1 + py + a: py = 0: dy {0} | {1, 2}; py = 1: {0, 1} | {2}."""
        w = self._w_ref                                        # [cout][cin][3][3], the layer's own layout
        cout, cin = w.shape[0], w.shape[1]
        rows = (((0,), (1, 2)), ((0, 1), (2,)))
        wsum = torch.empty(4, cout, cin, 4, device=w.device, dtype=torch.float32)
        for py in range(2):
            for px in range(2):
                for a_ in range(2):
                    for b_ in range(2):
                        acc = None
                        for dy in rows[py][a_]:
                            for dx in rows[px][b_]:
                                acc = w[:, :, dy, dx] if acc is None else acc + w[:, :, dy, dx]
                        wsum[2 * py + px, :, :, 2 * a_ + b_] = acc
        return wsum

    def h2_up(self):
        """(packing, scale) of the f16 x 2 form over a 2 x 2 nearest-up-sampled source (sda_pack_conv_weight_h2_up), or None: the four
        output parity classes as 2 x 2-tap convolutions with pre-summed taps.  Built on first use (warm-up, never under capture)."""
        if self.h2 is None or self._transpose or getattr(self, '_w_ref', None) is None:
            return None
        if getattr(self, '_h2_up', None) is None:
            lib = _lib.load()
            cout, cin = self._w_ref.shape[0], self._w_ref.shape[1]
            nbytes = int(lib.sda_conv_h2_up_packed_bytes(cout, cin))
            self._h2_up = False
            if nbytes > 0:
                wsum = self._h2_wsum()
                amax = float(wsum.abs().max())
                if amax > 0.0 and math.isfinite(amax):
                    buf = torch.empty(nbytes, device=wsum.device, dtype=torch.uint8)
                    _lib.check(lib.sda_pack_conv_weight_h2_up(wsum.data_ptr(), cout, cin, amax, buf.data_ptr(), _stream()),
                               'sda_pack_conv_weight_h2_up')
                    torch.cuda.current_stream(wsum.device).synchronize()      # (wsum is a temporary)
                    self._h2_up = (buf, float(lib.sda_conv_h2_scale(amax)))
        return self._h2_up or None

    def h2_pool(self):
        """(packing, scale) for the VJP of such a tail summed over the 2 x 2 up-sampling cells (this object is the layer's backward-data
        form): a 2 x 2-tap convolution over the four parity planes of the fine-resolution gradient (sda_pack_conv_weight_h2_rows), or None."""
        if self.h2 is None or not self._transpose or getattr(self, '_w_ref', None) is None:
            return None
        if getattr(self, '_h2_pool', None) is None:
            lib = _lib.load()
            cout, cin = self._w_ref.shape[0], self._w_ref.shape[1]
            nbytes = int(lib.sda_conv_h2_rows_packed_bytes(cin, 4 * cout, 4)) if self.m_real == cin else 0
            self._h2_pool = False
            if nbytes > 0:
                wrows = self._h2_wsum().permute(2, 0, 1, 3).reshape(cin, 4 * cout, 4).contiguous()      # [ci][class * cout + co][tap]
                amax = float(wrows.abs().max())
                if amax > 0.0 and math.isfinite(amax):
                    buf = torch.empty(nbytes, device=wrows.device, dtype=torch.uint8)
                    _lib.check(lib.sda_pack_conv_weight_h2_rows(wrows.data_ptr(), cin, 4 * cout, 4, amax, buf.data_ptr(), _stream()),
                               'sda_pack_conv_weight_h2_rows')
                    torch.cuda.current_stream(wrows.device).synchronize()
                    self._h2_pool = (buf, float(lib.sda_conv_h2_scale(amax)))
        return self._h2_pool or None

    def _h2_classes(self, vjp: bool):
        """(packing, scale) of a STRIDE-2 3 x 3 layer on conv_h2's per-class tap lists (csrc/conv_h2.hip MODE 3 / 4): class (py, px) has the
        taps dy in ((1,), (0, 2))[py], dx alike -- 1 / 2 / 2 / 4 of the layer's 9 -- packed class after class by
        sda_pack_conv_weight_h2_rows; vjp: rows = forward cin, K = forward cout (the input VJP), else the forward orientation."""
        lib = _lib.load()
        w = self._w_ref                                        # [cout][cin][3][3]
        cout, cin = w.shape[0], w.shape[1]
        rows, k = (cin, cout) if vjp else (cout, cin)
        sizes = [int(lib.sda_conv_h2_rows_packed_bytes(rows, k, (1 + (c >> 1)) * (1 + (c & 1)))) for c in range(4)]
        amax = float(w.abs().max())
        if min(sizes) <= 0 or not (amax > 0.0 and math.isfinite(amax)):
            return False
        buf = torch.empty(sum(sizes), device=w.device, dtype=torch.uint8)
        taps = ((1,), (0, 2))
        off = 0
        for c in range(4):
            wc = torch.stack([w[:, :, dy, dx] for dy in taps[c >> 1] for dx in taps[c & 1]], dim=-1)      # [cout][cin][nt]
            if vjp:
                wc = wc.permute(1, 0, 2)
            wc = wc.contiguous()
            _lib.check(lib.sda_pack_conv_weight_h2_rows(wc.data_ptr(), rows, k, wc.shape[-1], amax, buf[off:].data_ptr(), _stream()),
                       'sda_pack_conv_weight_h2_rows')
            torch.cuda.current_stream(w.device).synchronize()  # (wc is a temporary)
            off += sizes[c]
        return (buf, float(lib.sda_conv_h2_scale(amax)))

    def h2_zins(self):
        """The input VJP of a stride-2 3 x 3 layer (this object is its backward-data form) on conv_h2's parity-class form, or None."""
        if self.h2 is None or not self._transpose or getattr(self, '_w_ref', None) is None or self.m_real != self._w_ref.shape[1]:
            return None
        if getattr(self, '_h2_zins', None) is None:
            self._h2_zins = self._h2_classes(True)
        return self._h2_zins or None

    def h2_s2(self):
        """A stride-2 3 x 3 layer on conv_h2's parity-plane form (forward), or None."""
        if self.h2 is None or self._transpose or getattr(self, '_w_ref', None) is None:
            return None
        if getattr(self, '_h2_s2', None) is None:
            self._h2_s2 = self._h2_classes(False)
        return self._h2_s2 or None

    def wino4_zp(self) -> Optional[Tensor]:
        """The zero-position packing of `wino4` (None without it): what the up-sampling tails (sda/nn.py:161-169 of the reference) and
        their VJPs multiply with."""
        return self._wino4_zp


# ------------------------------------------------------------------------------------------ fused 1-D residual block

BLOCK1D = os.environ.get('SDA_BLOCK1D', '1') != '0'


def block1d_eligible(c: int, h: int, pk1: 'PackedConv', pk2: 'PackedConv') -> bool:
    """One-launch residual block (block1d.hip): 1-D, <= 64 channels, two k = 3 stride-1 convolutions c -> c."""
    return (BLOCK1D and h == 1 and 2 <= c <= 64 and (pk1.kh, pk1.kw) == (1, 3) and (pk2.kh, pk2.kw) == (1, 3) and
            pk1.k_pad <= 64 and pk1.m_pad <= 64 and pk2.k_pad == pk1.k_pad and pk2.m_pad == pk1.m_pad and
            pk1.k_real == c and pk1.m_real == c and pk2.k_real == c and pk2.m_real == c)


def _bracket_block1d(family: str, d, launch):
    """bench.py's roofline leg: a fused 1-D residual block is two k = 3 convolutions c -> c over n x len positions."""
    prof = conv_profile
    if prof is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    prof.records.append((e0, e1, 2 * (2.0 * d.n * d.len * d.c * d.c * 3), family))


def _block1d_desc(a, mod, mod_sn, pk1, pk2, circular, act, eps, unbiased):
    d = _lib.Block1dDesc()
    d.n, d.c, d.len = a.shape[0], a.shape[1], a.shape[-1]
    d.circular, d.act, d.unbiased, d.eps = int(circular), act, int(unbiased), eps
    d.k_pad, d.m_pad = pk1.k_pad, pk1.m_pad
    d.a = a.data_ptr()
    d.mod, d.mod_sn = _ptr(mod), mod_sn
    d.w1, d.w2 = pk1.packed.data_ptr(), pk2.packed.data_ptr()
    d.b1 = None if pk1.bias is None else pk1.bias.data_ptr()
    d.b2 = None if pk2.bias is None else pk2.bias.data_ptr()
    return d


def block1d_fwd(a: Tensor, mod, mod_sn: int, pk1: 'PackedConv', pk2: 'PackedConv', circular: bool, act: int, eps: float,
                unbiased: bool, y: Tensor, z: Optional[Tensor] = None, mean: Optional[Tensor] = None,
                rstd: Optional[Tensor] = None):
    """y = a + conv2(act(conv1(LN(a + mod)))) for planar a (n, c, 1, len); z / mean / rstd are kept for the VJP when given."""
    _dev(a, mod, y, z, mean, rstd)
    d = _block1d_desc(a, mod, mod_sn, pk1, pk2, circular, act, eps, unbiased)
    d.y, d.z, d.mean, d.rstd = y.data_ptr(), _ptr(z), _ptr(mean), _ptr(rstd)
    _bracket_block1d('block1d_fwd', d, lambda: _lib.check(_lib.load().sda_block1d_fwd(ctypes.byref(d), _stream()), 'sda_block1d_fwd'))


def block1d_bwd(g: Tensor, a: Tensor, z: Tensor, mean: Tensor, rstd: Tensor, mod, mod_sn: int, pk1b: 'PackedConv',
                pk2b: 'PackedConv', circular: bool, act: int, unbiased: bool, gx: Tensor):
    """gx = g + LN^T(conv1^T(act'(z) . conv2^T(g))); pk1b / pk2b: the backward-data packings."""
    _dev(g, a, z, mean, rstd, mod, gx)
    d = _block1d_desc(a, mod, mod_sn, pk1b, pk2b, circular, act, 0.0, unbiased)
    d.z, d.mean, d.rstd, d.g, d.gx = z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), gx.data_ptr()
    _bracket_block1d('block1d_bwd', d, lambda: _lib.check(_lib.load().sda_block1d_bwd(ctypes.byref(d), _stream()), 'sda_block1d_bwd'))


# ------------------------------------------------------------------------------------------ whole single-level 1-D net

NET1D = os.environ.get('SDA_NET1D', '1') != '0'


def net1d_launch(d: '_lib.Net1dDesc', backward: bool):
    """One launch for a whole single-level 1-D U-Net (csrc/net1d.hip), forward or input VJP; see include/sda_hip.h.
    (engine.net1d_plan mirrors the kernel's eligibility checks, so SDA_E_UNSUPPORTED here is a planning bug and raises.)"""
    lib = _lib.load()
    fn, name = (lib.sda_net1d_bwd, 'sda_net1d_bwd') if backward else (lib.sda_net1d_fwd, 'sda_net1d_fwd')
    prof = conv_profile
    if prof is None:
        _lib.check(fn(ctypes.byref(d), _stream()), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(fn(ctypes.byref(d), _stream()), name)
    e1.record()
    flops = 2.0 * d.n * d.len * 3 * (d.cin * d.c + 2 * d.nblocks * d.c * d.c + d.c * d.cout)
    prof.records.append((e0, e1, flops, 'net1d_bwd' if backward else 'net1d_fwd'))


def net1d_launch_fused(d: '_lib.Net1dDesc', f: '_lib.Net1dFuse', backward: bool):
    """One half of a fused guided evaluation (sda_net1d_fwd_fused / sda_net1d_bwd_fused; sda_amd/fused1d.py)."""
    lib = _lib.load()
    fn, name = (lib.sda_net1d_bwd_fused, 'sda_net1d_bwd_fused') if backward else (lib.sda_net1d_fwd_fused, 'sda_net1d_fwd_fused')
    prof = conv_profile
    if prof is None:
        _lib.check(fn(ctypes.byref(d), ctypes.byref(f), _stream()), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(fn(ctypes.byref(d), ctypes.byref(f), _stream()), name)
    e1.record()
    flops = 2.0 * d.n * d.len * 3 * (d.cin * d.c + 2 * d.nblocks * d.c * d.c + d.c * d.cout)
    prof.records.append((e0, e1, flops, 'net1d_bwd' if backward else 'net1d_fwd'))


# ------------------------------------------------------------------------------------------ LayerNorm pieces

def ln_stats(x: Tensor, mod: Optional[Tensor], mod_sn: int, eps: float, unbiased: bool, mean: Tensor, rstd: Tensor):
    """x: planar [n][c][hw] contiguous."""
    _dev(x, mod, mean, rstd)
    n, c = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()

    def launch():
        _lib.check(_lib.load().sda_ln_stats(x.data_ptr(), n, c, hw, _ptr(mod), mod_sn, eps, int(unbiased),
                                            mean.data_ptr(), rstd.data_ptr(), _stream()), 'sda_ln_stats')
    if conv_profile is not None:
        conv_profile.bracket_mem('ln_stats', 4.0 * n * hw * (c + 2), launch)       # read x once, write mean + rstd
    else:
        launch()


def ln_apply(x: Tensor, mod: Optional[Tensor], mod_sn: int, mean: Tensor, rstd: Tensor, y: Tensor):
    _dev(x, mod, mean, rstd, y)
    n, c = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    _lib.check(_lib.load().sda_ln_apply(x.data_ptr(), n, c, hw, _ptr(mod), mod_sn, mean.data_ptr(), rstd.data_ptr(),
                                        y.data_ptr(), _stream()), 'sda_ln_apply')


def ln_bwd(gh: Tensor, x: Tensor, h: int, w: int, mod: Optional[Tensor], mod_sn: int, mean: Tensor, rstd: Tensor,
           unbiased: bool, pool, res: Optional[Tensor], gx: Tensor, out_amax: Optional[Tensor] = None):
    """pool: (pool_h, pool_w) -- the nearest-upsample factors whose backward (cell sums of gh) is fused in; (1, 1) = none.
    out_amax (device scalar, optional): receives max |gx| -- the input scale of an f16 x 2 convolution that reads gx next."""
    _dev(gh, x, mod, mean, rstd, res, gx, out_amax)
    n, c = x.shape[0], x.shape[1]

    def launch():
        if out_amax is None:
            _lib.check(_lib.load().sda_ln_bwd(gh.data_ptr(), x.data_ptr(), n, c, h, w, _ptr(mod), mod_sn, mean.data_ptr(),
                                              rstd.data_ptr(), int(unbiased), pool[0], pool[1], _ptr(res), gx.data_ptr(),
                                              _stream()), 'sda_ln_bwd')
        else:
            _lib.check(_lib.load().sda_ln_bwd_amax(gh.data_ptr(), x.data_ptr(), n, c, h, w, _ptr(mod), mod_sn, mean.data_ptr(),
                                                   rstd.data_ptr(), int(unbiased), pool[0], pool[1], _ptr(res), gx.data_ptr(),
                                                   out_amax.data_ptr(), _stream()), 'sda_ln_bwd_amax')
    if conv_profile is not None:
        # read gh (at the pooled resolution), x, res; write gx; + the statistics
        nb = 4.0 * n * h * w * (c * (pool[0] * pool[1] + 2 + (1 if res is not None else 0)) + 2)
        conv_profile.bracket_mem('ln_bwd', nb, launch)
    else:
        launch()


# ------------------------------------------------------------------------------------------ time embedding

def time_embed(t: Tensor, freqs: Tensor, w0: Tensor, b0: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    _dev(t, freqs, w0, b0, w2, b2)
    nt = t.numel()
    e = w2.shape[0]
    emb = torch.empty(nt, e, device=t.device, dtype=torch.float32)
    _lib.check(_lib.load().sda_time_embed(t.data_ptr(), nt, freqs.data_ptr(), freqs.numel(), w0.data_ptr(), b0.data_ptr(),
                                          w0.shape[0], w2.data_ptr(), b2.data_ptr(), e, emb.data_ptr(), _stream()),
               'sda_time_embed')
    return emb


def linear_small(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    _dev(x, w, b)
    rows, in_f = x.shape
    out_f = w.shape[0]
    y = torch.empty(rows, out_f, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().sda_linear_small(x.data_ptr(), rows, in_f, w.data_ptr(), b.data_ptr(), out_f, y.data_ptr(),
                                            _stream()), 'sda_linear_small')
    return y


# ------------------------------------------------------------------------------------------ ResMLP pieces

def linear(x: Tensor, w: Tensor, b: Optional[Tensor], *, trans_w: bool = False, act_in: int = 0, act_out: int = 0,
           dact_z: Optional[Tensor] = None, act_d: int = 0, res: Optional[Tensor] = None) -> Tensor:
    """x: (rows, in) contiguous.  trans_w=False: w is torch's [out][in] (forward);  True: w is [in][out] (gx = gy @ W)."""
    _dev(x, w, b, dact_z, res)
    rows, in_f = x.shape
    out_f = w.shape[1] if trans_w else w.shape[0]
    assert (w.shape[0] if trans_w else w.shape[1]) == in_f
    y = torch.empty(rows, out_f, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().sda_linear(x.data_ptr(), rows, in_f, w.data_ptr(), _ptr(b), out_f, int(trans_w), act_in,
                                      act_out, _ptr(dact_z), act_d, _ptr(res), y.data_ptr(), _stream()), 'sda_linear')
    return y


def row_ln(x: Tensor, eps: float, unbiased: bool, y: Tensor, mean: Optional[Tensor] = None,
           rstd: Optional[Tensor] = None):
    _dev(x, y, mean, rstd)
    rows, f = x.shape
    _lib.check(_lib.load().sda_row_ln(x.data_ptr(), rows, f, eps, int(unbiased), y.data_ptr(), _ptr(mean), _ptr(rstd),
                                      _stream()), 'sda_row_ln')


def row_ln_bwd(gh: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, unbiased: bool, res: Optional[Tensor], gx: Tensor):
    _dev(gh, x, mean, rstd, res, gx)
    rows, f = x.shape
    _lib.check(_lib.load().sda_row_ln_bwd(gh.data_ptr(), x.data_ptr(), rows, f, mean.data_ptr(), rstd.data_ptr(),
                                          int(unbiased), _ptr(res), gx.data_ptr(), _stream()), 'sda_row_ln_bwd')


def mlp_launch(d: '_lib.MlpDesc', backward: bool):
    """A whole ResMLP in one launch (csrc/mlp1d.hip), forward or input VJP; see include/sda_hip.h."""
    lib = _lib.load()
    fn, name = (lib.sda_mlp_bwd, 'sda_mlp_bwd') if backward else (lib.sda_mlp_fwd, 'sda_mlp_fwd')
    prof = conv_profile
    if prof is None:
        _lib.check(fn(ctypes.byref(d), _stream()), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(fn(ctypes.byref(d), _stream()), name)
    e1.record()
    flops = 2.0 * d.rows * sum(d.in_f[g] * d.out_f[g] for g in range(d.ngemm))
    prof.records.append((e0, e1, flops, 'mlp_bwd' if backward else 'mlp_fwd'))


def mlp_launch_win(d: '_lib.MlpDesc', w: '_lib.MlpWin', backward: bool):
    """One half of a fused guided evaluation of a local score network (sda_mlp_fwd_win / sda_mlp_bwd_win; sda_amd/fused1d.py)."""
    lib = _lib.load()
    fn, name = (lib.sda_mlp_bwd_win, 'sda_mlp_bwd_win') if backward else (lib.sda_mlp_fwd_win, 'sda_mlp_fwd_win')
    prof = conv_profile
    if prof is None:
        _lib.check(fn(ctypes.byref(d), ctypes.byref(w), _stream()), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(fn(ctypes.byref(d), ctypes.byref(w), _stream()), name)
    e1.record()
    flops = 2.0 * d.rows * sum(d.in_f[g] * d.out_f[g] for g in range(d.ngemm))
    prof.records.append((e0, e1, flops, 'mlp_bwd' if backward else 'mlp_fwd'))


def mc_finish(eps: Tensor, ghat: Tensor, gwin: Tensor, b: int, nw: int, k: int, c: int, cx0: float, cx1: float, coef_ptr: int, mode: int,
              out: Optional[Tensor], xs: Optional[Tensor], step_coef_ptr: int, partial: Optional[Tensor]):
    _dev(eps, ghat, gwin, out, xs, partial)
    _lib.check(_lib.load().sda_mc_finish(eps.data_ptr(), ghat.data_ptr(), gwin.data_ptr(), b, nw, k, c, cx0, cx1, coef_ptr, mode,
                                         _ptr(out), _ptr(xs), step_coef_ptr, _ptr(partial), _stream()), 'sda_mc_finish')


# ------------------------------------------------------------------------------------------ fold / unfold adjoints

def fold(s: Tensor, b: int, nw: int, k: int, c: int, hw: int, out: Tensor):
    _dev(s, out)
    _lib.check(_lib.load().sda_fold(s.data_ptr(), b, nw, k, c, hw, out.data_ptr(), _stream()), 'sda_fold')


def fold_adjoint(g_out: Tensor, b: int, nw: int, k: int, c: int, hw: int, g_s: Tensor):
    _dev(g_out, g_s)
    _lib.check(_lib.load().sda_fold_adjoint(g_out.data_ptr(), b, nw, k, c, hw, g_s.data_ptr(), _stream()),
               'sda_fold_adjoint')


def unfold_adjoint(g_win: Tensor, b: int, nw: int, k: int, c: int, hw: int, win_c_total: int, g_x: Tensor):
    _dev(g_win, g_x)
    _lib.check(_lib.load().sda_unfold_adjoint(g_win.data_ptr(), b, nw, k, c, hw, win_c_total, g_x.data_ptr(), _stream()),
               'sda_unfold_adjoint')


# ------------------------------------------------------------------------------------------ PC updates / guidance

def pc_predict(x: Tensor, eps: Tensor, r: float, c1: float, coef_dev: Optional[Tensor] = None):
    _dev(x, eps, coef_dev)
    _lib.check(_lib.load().sda_pc_predict(x.data_ptr(), eps.data_ptr(), x.numel(), r, c1, _ptr(coef_dev), _stream()),
               'sda_pc_predict')


SUMSQ_CHUNKS = 64


def _check_partial(partial: Tensor, b: int):
    if partial.numel() < b * SUMSQ_CHUNKS or not partial.is_contiguous():
        raise _lib.SdaHipError(f'partial-sum buffer needs {b} x {SUMSQ_CHUNKS} contiguous floats, got {tuple(partial.shape)}')


def sumsq_chunks(per_sample: int) -> int:
    """Workgroups per sample of the two-stage sum of squares: one per 4096 elements, at most SUMSQ_CHUNKS.  (A fixed 64 gave the Lorenz
    trajectories -- 195 elements -- 64 workgroups of three elements each: 32 us per correction at eval.py's batch.)"""
    return max(1, min(SUMSQ_CHUNKS, per_sample // 4096))


def sumsq_partial(eps: Tensor, b: int, partial: Tensor) -> int:
    """partial[b][nchunk] <- per-chunk sums of eps^2; returns nchunk (pass it to pc_correct)."""
    _dev(eps, partial)
    _check_partial(partial, b)
    per = eps.numel() // b
    nchunk = sumsq_chunks(per)
    _lib.check(_lib.load().sda_sumsq_partial(eps.data_ptr(), b, per, partial.data_ptr(), nchunk, _stream()),
               'sda_sumsq_partial')
    return nchunk


def pc_correct(x: Tensor, eps: Tensor, z: Tensor, b: int, partial: Tensor, tau: float, sigma: float,
               coef_dev: Optional[Tensor] = None, nchunk: Optional[int] = None):
    _dev(x, eps, z, partial, coef_dev)
    if nchunk is None:
        nchunk = sumsq_chunks(x.numel() // b)             # (what sumsq_partial wrote for this sample size)
    if partial.numel() < b * nchunk or not partial.is_contiguous():
        raise _lib.SdaHipError(f'partial-sum buffer needs {b} x {nchunk} contiguous floats, got {tuple(partial.shape)}')
    per = x.numel() // b
    _lib.check(_lib.load().sda_pc_correct(x.data_ptr(), eps.data_ptr(), z.data_ptr(), b, per, partial.data_ptr(),
                                          nchunk, tau, sigma, _ptr(coef_dev), _stream()), 'sda_pc_correct')


#: rows sda_pc_correct_keyed takes (its batch axis is grid.y; csrc/step1d.hip returns SDA_E_BADARG above)
PC_KEYED_MAX_ROWS = 65535


def pc_correct_keyed(x: Tensor, eps: Tensor, b: int, partial: Tensor, nchunk: int, tau: float, coef_dev: Tensor, seed: int, row0: int,
                     draw_dev: Tensor, draw_mul: int, draw_add: int):
    """The corrector update with its row-keyed noise generated in the kernel (the z of randn_rows(seed, row0, draw_dev * mul + add))."""
    _dev(x, eps, partial, coef_dev)
    if partial.numel() < b * nchunk or not partial.is_contiguous():
        raise _lib.SdaHipError(f'partial-sum buffer needs {b} x {nchunk} contiguous floats, got {tuple(partial.shape)}')
    if not draw_dev.is_cuda or draw_dev.dtype != torch.int64:
        raise _lib.SdaHipError('draw_dev must be a device int64 scalar')
    per = x.numel() // b
    _lib.check(_lib.load().sda_pc_correct_keyed(x.data_ptr(), eps.data_ptr(), b, per, partial.data_ptr(), nchunk, tau, 0.0,
                                                coef_dev.data_ptr(), seed & 0xffffffffffffffff, row0, draw_dev.data_ptr(), draw_mul,
                                                draw_add, _stream()), 'sda_pc_correct_keyed')


def randn_rows(out: Tensor, seed: int, row0: int, draw: int = 0, draw_dev: Optional[Tensor] = None, draw_mul: int = 1,
               draw_add: int = 0):
    """out (rows, ...) <- N(0, 1) draws keyed per row: out[r] depends on (seed, row0 + r, draw) only.  With ``draw_dev`` (a
    device int64 scalar) the draw index is ``draw_dev * draw_mul + draw_add``, read on the device (graph replay)."""
    _dev(out)
    if not out.is_contiguous():
        raise _lib.SdaHipError('randn_rows writes a contiguous tensor')
    if draw_dev is not None and (not draw_dev.is_cuda or draw_dev.dtype != torch.int64):
        raise _lib.SdaHipError('draw_dev must be a device int64 scalar')
    rows = out.shape[0]
    if out.numel() == 0:                    # an empty shard (batch < world size): nothing to draw
        return out
    _lib.check(_lib.load().sda_randn_rows(out.data_ptr(), rows, out.numel() // max(rows, 1), seed & 0xffffffffffffffff, row0,
                                          draw, _ptr(draw_dev), draw_mul, draw_add, _stream()), 'sda_randn_rows')
    return out


def vp_schedule(t: Tensor, alpha_kind: int, eta: float, k: float, sigma_kind: int) -> Tensor:
    """Device scalar t -> device pair {mu(t), sigma(t)} in one launch."""
    _dev(t)
    out = torch.empty(2, device=t.device, dtype=torch.float32)
    _lib.check(_lib.load().sda_vp_schedule(t.data_ptr(), alpha_kind, eta, k, sigma_kind, out.data_ptr(), _stream()),
               'sda_vp_schedule')
    return out


def _adjacent_pair(mu, sigma):
    """mu, sigma that are elements 0 and 1 of one fp32 device buffer (what vp_schedule returns) already form the pair."""
    if isinstance(mu, Tensor) and isinstance(sigma, Tensor) and mu.is_cuda and mu.dtype == sigma.dtype == torch.float32 \
            and mu.numel() == sigma.numel() == 1 and sigma.data_ptr() == mu.data_ptr() + 4 \
            and mu.untyped_storage().data_ptr() == sigma.untyped_storage().data_ptr():
        return mu.as_strided((2,), (1,))
    return None


def _coef(mu, sigma):
    """python floats travel by value; 0-dim device tensors travel as a device {mu, sigma} pair (no host sync)."""
    if isinstance(mu, Tensor) or isinstance(sigma, Tensor):
        pair = _adjacent_pair(mu, sigma)
        if pair is not None:
            return 0.0, 0.0, pair
        dev = mu.device if isinstance(mu, Tensor) else sigma.device
        pair = torch.stack([torch.as_tensor(mu, dtype=torch.float32, device=dev).reshape(()),
                            torch.as_tensor(sigma, dtype=torch.float32, device=dev).reshape(())])
        return 0.0, 0.0, pair
    return float(mu), float(sigma), None


def denoise(x: Tensor, eps: Tensor, mu, sigma, xhat: Tensor):
    _dev(x, eps, xhat)
    m, s, pair = _coef(mu, sigma)
    _lib.check(_lib.load().sda_denoise(x.data_ptr(), eps.data_ptr(), x.numel(), m, s, _ptr(pair), xhat.data_ptr(),
                                       _stream()), 'sda_denoise')


def guided_combine(eps: Tensor, ghat: Tensor, vjp: Optional[Tensor], mu, sigma, out: Tensor):
    _dev(eps, ghat, vjp, out)
    m, s, pair = _coef(mu, sigma)
    _lib.check(_lib.load().sda_guided_combine(eps.data_ptr(), ghat.data_ptr(), _ptr(vjp), eps.numel(), m, s, _ptr(pair),
                                              out.data_ptr(), _stream()), 'sda_guided_combine')


def gauss_cotangent(y: Tensor, ax: Tensor, std: float, gamma: float, mu, sigma) -> Tensor:
    """(y - ax) / (std^2 + gamma (sigma/mu)^2) for scalar std / gamma; y broadcasts over ax's leading axis."""
    _dev(y, ax)
    y, ax = y.contiguous(), ax.contiguous()
    out = torch.empty_like(ax)
    m, s, pair = _coef(mu, sigma)
    _lib.check(_lib.load().sda_gauss_cotangent(y.data_ptr(), y.numel(), ax.data_ptr(), ax.numel(), float(std), float(gamma), m, s,
                                               _ptr(pair), out.data_ptr(), _stream()), 'sda_gauss_cotangent')
    return out


# ---------------------------------------------------------------------------------------------- evaluation metrics
def pairwise_dist(x: Tensor, y: Tensor, squared: bool) -> Tensor:
    """x: (m, d), y: (n, d) -> (m, n) Euclidean distances (or their squares)."""
    _dev(x, y)
    x, y = x.contiguous(), y.contiguous()
    if x.dim() != 2 or y.dim() != 2 or x.shape[1] != y.shape[1]:
        raise _lib.SdaHipError(f'pairwise_dist: incompatible shapes {tuple(x.shape)} / {tuple(y.shape)}')
    out = torch.empty(x.shape[0], y.shape[0], device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().sda_pairwise_dist(x.data_ptr(), x.shape[0], y.data_ptr(), y.shape[0], x.shape[1],
                                             0 if squared else 1, out.data_ptr(), _stream()), 'sda_pairwise_dist')
    return out


def mmd_kernel_sum(d2: Tensor) -> Tensor:
    """sum_ij sum_sigma exp(-d2_ij / sigma) as a 0-dim float64 device tensor."""
    _dev(d2)
    d2 = d2.contiguous()
    nblocks = int(min(1024, max(1, (d2.numel() + 255) // 256)))
    partial = torch.empty(nblocks, device=d2.device, dtype=torch.float64)
    _lib.check(_lib.load().sda_mmd_kernel_sums(d2.data_ptr(), d2.numel(), partial.data_ptr(), nblocks, _stream()),
               'sda_mmd_kernel_sums')
    return partial.sum()


def transport_cost(cost: Tensor) -> float:
    """Host optimal-transport solve with uniform marginals of an (m, n) fp32 host cost matrix: min_P <P, cost>."""
    if cost.is_cuda or cost.dtype != torch.float32 or cost.dim() != 2:
        raise _lib.SdaHipError('transport_cost expects an fp32 host matrix')
    cost = cost.contiguous()
    total = ctypes.c_double(0.0)
    _lib.check(_lib.load().sda_transport_cost(cost.data_ptr(), cost.shape[0], cost.shape[1], ctypes.addressof(total)),
               'sda_transport_cost')
    return total.value


def assignment_cost(cost: Tensor):
    """Host linear-assignment solve of a square cost matrix (CPU tensor): (minimum total cost, column of each row)."""
    if cost.is_cuda or cost.dtype != torch.float32 or cost.dim() != 2 or cost.shape[0] != cost.shape[1]:
        raise _lib.SdaHipError('assignment_cost expects a square fp32 host matrix')
    cost = cost.contiguous()
    n = cost.shape[0]
    total = ctypes.c_double(0.0)
    cols = torch.empty(n, dtype=torch.int32)
    _lib.check(_lib.load().sda_assignment_cost(cost.data_ptr(), n, ctypes.addressof(total), cols.data_ptr()),
               'sda_assignment_cost')
    return total.value, cols


def clock_probe(device, ms: float = 2.0, blocks: int = 1024) -> dict:
    """The shader clock (GHz) the fp32 matrix-core stream sustains on ``device`` right now (``sda_clock_probe``): ``blocks``
    workgroups of four waves issue register-resident v_mfma_f32_16x16x4_f32 for about ``ms`` milliseconds and time themselves with
    the shader-clock and the 100 MHz counters.  Measurement support for bench.py; synchronises the device."""
    lib = _lib.load()
    out = torch.zeros(2 * blocks, device=device, dtype=torch.int64)
    sink = torch.zeros(1, device=device, dtype=torch.float32)
    # 8 MFMAs x 32 cycles per round and wave, one wave per SIMD per resident workgroup; `blocks` over 256 CUs run in waves of 256 x
    # (workgroups per CU): size the rounds so that the whole launch lasts ~ms at 2.4 GHz
    per_cu = max(1, -(-blocks // 256))
    iters = max(64, int(ms * 1e-3 * 2.4e9 / (8 * 32) / per_cu))
    with torch.cuda.device(device):
        for _ in range(2):                                  # (the first launch also pays the code upload)
            _lib.check(lib.sda_clock_probe(out.data_ptr(), blocks, sink.data_ptr(), iters, _stream()), 'sda_clock_probe')
        torch.cuda.synchronize(device)
    v = out.reshape(blocks, 2).double()
    ghz = (v[:, 0] / v[:, 1].clamp_min(1)) * 0.1
    return {'ghz': float(ghz.median()), 'ghz_min': float(ghz.min()), 'ghz_max': float(ghz.max()), 'blocks': blocks,
            'mfma_per_wave': iters * 8, 'instruction': 'v_mfma_f32_16x16x4_f32, register operands'}
