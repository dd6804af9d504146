"""Executor of the modulated residual U-Net on MI355X: forward and backward-data (VJP w.r.t. the input).

It sequences the gfx950 kernels of libsda_hip.so for the reference's ``UNet.forward`` (sda/nn.py:184-206) and for
the gradient torch.autograd would propagate through it at sda/score.py:394 (guidance needs d/dx only -- no weight
gradients are ever formed).

Layout: every internal activation is PLANAR ``[n][c][h][w]`` fp32 (1-D nets: ``h = 1``).  Per residual block the
forward runs three kernels and materialises two tensors:

    stats(a)                     -> mean, rstd              (channel LayerNorm statistics of a + mod)
    conv1(LN(a + mod)) + b1      -> z                       (LN + modulation applied inside the conv loader)
    conv2(act(z)) + b2 + a       -> a'                      (activation in the loader, residual in the epilogue)

so only ``a`` (block input) and ``z`` (pre-activation) are saved for the VJP, which is again three kernels:

    conv2^T(g) * act'(z)         -> gz
    conv1^T(gz)                  -> gh
    g + LN_bwd(gh; a, mod, mean, rstd) -> g'

The image axis is processed in chunks sized from free HBM (288 GB on MI355X); when a VJP is requested for more
images than can keep their activations resident, the forward is recomputed chunk by chunk inside the backward
("activation memory under guidance").
"""
import math
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from ._lib import SdaHipError
from .ops import PackedConv, conv_out_size, make_conv_desc


def _lib_net1d_maxb() -> int:
    from ._lib import NET1D_MAXB
    return NET1D_MAXB


def _net1d_span_ok(sc: int, sx: int, channels: int, length: int) -> bool:
    """(channel, position) strides the whole-net 1-D kernel can address: non-negative, one image spans < 2^30 elements."""
    return sc >= 0 and sx >= 0 and sc * channels + sx * length < 1 << 30


# fraction of currently-unallocated HBM one chunk's activations may occupy
CHUNK_HBM_FRACTION = 0.45
# fraction of unallocated HBM that activations kept from the forward for the VJP may occupy
KEEP_HBM_FRACTION = 0.45
# VJP of the stride-2 level heads as one small convolution per output parity class (False: one zero-insertion launch)
PARITY_SPLIT = True
_warned_training = False


@dataclass
class Source:
    """How the first convolution reads its input: a strided view (possibly the sliding-window view of a
    trajectory, score.py:146-153) plus optional broadcast context channels (score.py:87)."""
    x: Tensor
    n: int
    cx: int
    hs: int
    ws: int
    sn_outer: int
    sc: int
    sy: int
    sx: int
    sn_inner: int = 0
    n_inner: int = 1
    ctx: Optional[Tensor] = None
    cctx: int = 0
    ctx_sn: int = 0


def planar_source(a: Tensor) -> dict:
    n, c, h, w = a.shape
    return dict(x_ptr=a.data_ptr(), n=n, cx=c, hs=h, ws=w, x_sn_outer=c * h * w, x_sc=h * w, x_sy=w, x_sx=1)


class _ConvCache:
    """Packed (forward / backward-data) weights of one nn.ConvNd, rebuilt when the parameter changes."""

    def __init__(self, conv: nn.Module):
        if conv.dilation != (1,) * len(conv.dilation) or conv.groups != 1:
            raise NotImplementedError('dilated / grouped convolutions have no gfx950 kernel')
        if conv.padding_mode not in ('zeros', 'circular'):
            raise NotImplementedError(f"padding_mode={conv.padding_mode!r} (supported: 'zeros', 'circular')")
        ks = conv.kernel_size
        if any(k % 2 == 0 for k in ks):
            raise NotImplementedError('even kernel sizes are not supported')
        self.conv = conv
        self.kh, self.kw = (1, ks[0]) if len(ks) == 1 else ks
        st = conv.stride
        self.sh, self.sw = (1, st[0]) if len(st) == 1 else st
        self.circular = conv.padding_mode == 'circular'
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self._key = None
        self._fwd = None
        self._bwd = {}
        self._parity = None

    def _sync(self):
        # (data_ptr, _version) catches optimiser steps, .to(), load_state_dict and in-place ops on the parameter; writes
        # through `.data` (EMA swaps: p.data.copy_(ema)) bump neither -- those callers use UNet.invalidate_engine()
        w = self.conv.weight
        key = (w.data_ptr(), ops.tensor_version(w), str(w.device), None if self.conv.bias is None else ops.tensor_version(self.conv.bias),
               ops.MULTIPLY)                            # (switching the multiply re-packs: the f16 x 2 packing is made with the others)
        if key != self._key:
            self._key, self._fwd, self._bwd, self._parity = key, None, {}, None

    def invalidate(self):
        self._key = None

    def fwd(self) -> PackedConv:
        self._sync()
        if self._fwd is None:
            self._fwd = PackedConv(self.conv.weight, self.conv.bias, transpose=False)
        return self._fwd

    def bwd(self, cin_keep: Optional[int] = None) -> PackedConv:
        self._sync()
        if cin_keep not in self._bwd:
            self._bwd[cin_keep] = PackedConv(self.conv.weight, None, transpose=True, cin_keep=cin_keep)
        return self._bwd[cin_keep]

    def bwd_parity(self):
        """Backward-data form of a stride-2 (per axis) 3-tap convolution, split by output parity.

        g_x[2i + p] only receives the taps t with (p + 1 - t) even: t = 1 for p = 0 (from g[i]) and t = 0, 2 for p = 1
        (from g[i + 1], g[i]).  So each parity class is a small stride-1 convolution over g (1 or 2 taps per split axis,
        pad 0) writing an interleaved quarter (half, for 1-D) of g_x -- together 9 of the 36 multiplies the zero-insertion
        formulation spends per output pixel quad.  Returns [(py, px, PackedConv, (pad_h, pad_w))] or None when the layer
        does not have that shape (then the zero-insertion path is used)."""
        self._sync()
        if self._parity is None:
            taps = []
            for k, s in ((self.kh, self.sh), (self.kw, self.sw)):
                if s == 1:
                    taps.append([(None, list(range(k)), k // 2)])
                elif s == 2 and k == 3:
                    taps.append([(0, [1], 0), (1, [0, 2], 0)])
                else:
                    taps = None
                    break
            if taps is None or (self.sh == 1 and self.sw == 1):
                self._parity = False
            else:
                w = self.conv.weight.detach()
                if w.dim() == 3:
                    w = w.unsqueeze(2)
                out = []
                for py, ty, ph in taps[0]:
                    for px, tx, pw in taps[1]:
                        sub = w[:, :, ty][:, :, :, tx].contiguous()
                        out.append((py, px, PackedConv(sub, None, transpose=True), (ph, pw)))
                self._parity = out
        return self._parity or None

    def bwd_parity4(self):
        """The four parity packings back to back, [9 taps][k_pad][m_pad] in class order (0,0), (0,1), (1,0), (1,1), for the
        one-launch kernel (csrc/conv_par4.hip); None when the layer is not a 2-D stride-(2, 2) 3 x 3 convolution."""
        classes = self.bwd_parity()
        if not classes or len(classes) != 4 or [(py, px) for py, px, _, _ in classes] != [(0, 0), (0, 1), (1, 0), (1, 1)]:
            return None
        if getattr(self, '_parity4', None) is None or self._parity4[0] is not classes:
            pks = [pk for _, _, pk, _ in classes]
            if any(pk.k_pad != pks[0].k_pad or pk.m_pad != pks[0].m_pad or pk.m_pad != pk.m_real for pk in pks):
                return None
            self._parity4 = (classes, torch.cat([pk.packed for pk in pks]))
        return self._parity4[1]


def launch_conv(pk: PackedConv, src: dict, out: Tensor, ho: int, wo: int, *, circular: bool, stride=(1, 1), up=(1, 1),
                zins=(1, 1), bias: Optional[Tensor] = None, mod: Optional[Tensor] = None, mod_sn: int = 0,
                ln=None, act_in: int = 0, dact_z: Optional[Tensor] = None, act_d: int = 0, res: Optional[Tensor] = None,
                ctx: Optional[Tensor] = None, cctx: int = 0, ctx_sn: int = 0, pad=None, parity4_w: Optional[Tensor] = None,
                pool=(1, 1), x_amax=None, out_amax: Optional[Tensor] = None, h2_only: bool = False):
    """h2_only: launch only if the f16 x 2 kernel serves the descriptor, else return None WITHOUT launching (the caller has a faster
    fp32 form than the general kernel: the stride-2 head's VJP).
    pool != (1, 1): `out` is the pooled tensor [n][cout][ho / pool_h][wo / pool_w] (cell sums; sda_conv_desc.pool_h / pool_w);
    returns None when no kernel serves the pooled form (the caller runs the plain launch and pools in its reader)."""
    out_strides = (0, 0, 0, 0)
    if not out.is_contiguous():                      # an interleaved view of the real output (parity-split VJP)
        out_strides = tuple(out.stride())
        for t in (dact_z, res):
            if t is not None and tuple(t.stride()) != out_strides:
                raise SdaHipError('conv epilogue operands must share the layout of the output view')
    zp = pk.wino4_zp() if (tuple(up) == (2, 2) or tuple(pool) == (2, 2)) and hasattr(pk, 'wino4_zp') else None
    d = make_conv_desc(**src, w_ptr=pk.packed.data_ptr(), cin_pad=pk.k_pad, cout_pad=pk.m_pad, cout=pk.m_real,
                       kh=pk.kh, kw=pk.kw, out_ptr=out.data_ptr(), ho=ho, wo=wo, mt=pk.mt,
                       stride_h=stride[0], stride_w=stride[1], circular=circular, up_h=up[0], up_w=up[1],
                       zins_h=zins[0], zins_w=zins[1],
                       ctx_ptr=None if ctx is None else ctx.data_ptr(), cctx=cctx, ctx_sn=ctx_sn,
                       mod_ptr=None if mod is None else mod.data_ptr(), mod_sn=mod_sn,
                       ln_mean_ptr=None if ln is None else ln[0].data_ptr(),
                       ln_rstd_ptr=None if ln is None else ln[1].data_ptr(),
                       act_in=act_in, bias_ptr=None if bias is None else bias.data_ptr(),
                       dact_z_ptr=None if dact_z is None else dact_z.data_ptr(), act_d=act_d,
                       res_ptr=None if res is None else res.data_ptr(),
                       w_wino_ptr=None if getattr(pk, 'wino', None) is None else pk.wino.data_ptr(),
                       w_wino4_ptr=None if getattr(pk, 'wino4', None) is None else pk.wino4.data_ptr(),
                       pad=pad, out_strides=out_strides, pool=pool, w_wino4_zp_ptr=None if zp is None else zp.data_ptr())
    if pool != (1, 1):
        if tuple(pool) == (2, 2) and ops.H2_POOL and x_amax is not None and getattr(pk, 'h2', None) is not None:
            packing = pk.h2_pool()
            if packing is not None and ops.conv_h2(d, pk, x_amax, out_amax, packing=packing):
                return d
        return d if ops.conv_pooled(d) else None
    if getattr(pk, 'h2', None) is not None and parity4_w is None:
        # OPT-IN f16 x 2 multiply (ops.MULTIPLY): behind a LayerNorm the loader's output is bounded by sqrt(channels); otherwise the
        # caller says how large the input is (the producing launch's out_amax, or an ops.absmax pass)
        bound = math.sqrt(src['cx']) if ln is not None else x_amax
        packing = None
        if tuple(up) == (2, 2):                          # the tails: four parity classes of 2 x 2 pre-summed taps (PackedConv.h2_up)
            packing = pk.h2_up() if ops.H2_UP else None
            if packing is None:
                bound = None                             # -> the zero-position Winograd kernel
        elif tuple(zins) == (2, 2) or tuple(stride) == (2, 2):      # the stride-2 heads / their VJP: per-class tap lists
            packing = (pk.h2_zins() if tuple(zins) == (2, 2) else pk.h2_s2()) if ops.H2_S2 else None
            if packing is None:
                bound = None                             # -> the fp32 direct kernels
        if ops.conv_h2(d, pk, bound, out_amax, packing=packing):
            return d
    if h2_only:
        return None
    if parity4_w is not None:
        # all four parity classes of a stride-2 VJP in one launch: the class-(0,0) descriptor with the concatenated packing
        d.w = parity4_w.data_ptr()
        d.kh = d.kw = 2                                  # (the window all four classes share)
        return d if ops.conv_parity4(d) else None
    ops.conv_igemm(d)
    return d


class _Block:
    def __init__(self, block, width: int, mod_off: int):
        from .nn import LayerNorm, activation_id
        residue = block.residue
        if not (len(residue) == 4 and isinstance(residue[0], LayerNorm)):
            raise NotImplementedError('unexpected residue structure')
        self.ln = residue[0]
        self.conv1 = _ConvCache(residue[1])
        self.act = activation_id(residue[2])
        self.conv2 = _ConvCache(residue[3])
        self.project = block.project[0]
        self.width = width
        self.mod_off = mod_off


class _Level:
    pass


class UNetEngine:
    def __init__(self, unet):
        from .nn import LN_UNBIASED
        if unet.spatial not in (1, 2):
            raise NotImplementedError('this engine serves spatial = 1, 2 (spatial = 3: engine3d.UNet3dEngine)')
        self.unet = unet
        self.unbiased = LN_UNBIASED
        self.depth = len(unet.hidden_blocks)
        D = self.depth
        self.levels: List[_Level] = []
        off = 0
        for lvl in range(D):
            lev = _Level()
            lev.C = unet.hidden_channels[lvl]
            j = D - 1 - lvl                                  # tails / ascent are stored deepest first (nn.py:179-182)
            head = unet.heads[lvl] if lvl == 0 else unet.heads[lvl][0]
            lev.head = _ConvCache(head)
            if lvl == 0:
                lev.tail = _ConvCache(unet.tails[j])
                lev.tail_ln = None
            else:
                lev.tail_ln = unet.tails[j][0]
                lev.tail = _ConvCache(unet.tails[j][2])
            lev.descent, lev.ascent = [], []
            for blk in unet.descent[lvl]:
                lev.descent.append(_Block(blk, lev.C, off)); off += lev.C
            for blk in unet.ascent[j]:
                lev.ascent.append(_Block(blk, lev.C, off)); off += lev.C
            self.levels.append(lev)
        self.mod_total = off
        self._proj_key = None
        self._proj = None
        st = unet.stride
        self.sh, self.sw = (1, st[0]) if len(st) == 1 else st

    def invalidate(self):
        """Drop every packed-weight cache (forward / backward-data / Winograd / parity packs, the concatenated projection
        matrix).  The caches re-key themselves on parameter pointer and version, which in-place writes through ``.data``
        (``p.data.copy_(ema)``) do not change: call this after such a write."""
        for lev in self.levels:
            lev.head.invalidate()
            lev.tail.invalidate()
            for blk in lev.descent + lev.ascent:
                blk.conv1.invalidate()
                blk.conv2.invalidate()
        self._proj_key = None
        self.__dict__.pop('_net1d_cache', None)

    # -------------------------------------------------------------------------------- modulation vectors
    def _blocks(self):
        for lev in self.levels:
            yield from lev.descent
            yield from lev.ascent

    def projection(self):
        """All blocks' ``project`` Linears concatenated row-wise: one small kernel gives every modulation vector."""
        key = tuple((b.project.weight.data_ptr(), ops.tensor_version(b.project.weight), ops.tensor_version(b.project.bias)) for b in self._blocks())
        if key != self._proj_key:
            blocks = sorted(self._blocks(), key=lambda b: b.mod_off)
            w = torch.cat([b.project.weight.detach() for b in blocks], dim=0).contiguous()
            bias = torch.cat([b.project.bias.detach() for b in blocks], dim=0).contiguous()
            self._proj_key, self._proj = key, (w, bias)
        return self._proj

    def modulation(self, emb: Tensor) -> Tensor:
        """emb: (T, mod_features) -> (T, mod_total)."""
        w, b = self.projection()
        return ops.linear_small(emb.contiguous(), w, b)

    # -------------------------------------------------------------------------------- memory planning
    def bytes_per_image(self, hs: int, ws: int, save: bool) -> int:
        total, h, w = 0, hs, ws
        for lvl, lev in enumerate(self.levels):
            if lvl > 0:
                h = conv_out_size(h, lev.head.kh, lev.head.sh)
                w = conv_out_size(w, lev.head.kw, lev.head.sw)
            plane = lev.C * h * w * 4
            nblk = len(lev.descent) + len(lev.ascent)
            total += plane * ((2 * nblk + 4) if save else 5)
        return total

    def chunk_size(self, n: int, hs: int, ws: int, save: bool, device, fraction: Optional[float] = None) -> int:
        total = torch.cuda.get_device_properties(device).total_memory
        avail = max(total - torch.cuda.memory_allocated(device), total // 8)
        per = max(self.bytes_per_image(hs, ws, save), 1)
        return int(max(1, min(n, (avail * (fraction or CHUNK_HBM_FRACTION)) // per)))

    # -------------------------------------------------------------------------------- whole-batch drivers
    def forward_all(self, src: Source, mod_all, per_image: bool, out: Tensor, need_grad: bool):
        """All images of ``src`` -> ``out``; returns the VJP state: a list of (lo, hi, saved-or-None).

        With ``need_grad`` the activations of as many images as fit ``KEEP_HBM_FRACTION`` of the unallocated HBM are kept
        (all of them when they fit: one forward, one backward); the rest is recomputed chunk by chunk in the backward."""
        n = src.n
        dev = out.device
        state = []
        lo = 0
        if need_grad:
            keep = self.chunk_size(n, src.hs, src.ws, True, dev, KEEP_HBM_FRACTION)
            if keep >= n:
                state.append((0, n, self.forward_chunk(src, 0, n, mod_all, per_image, out, True)))
                return state
            if keep >= max(8, n // 16):                 # worth keeping a leading part
                state.append((0, keep, self.forward_chunk(src, 0, keep, mod_all, per_image, out[:keep], True)))
                lo = keep
        chunk = self.chunk_size(n - lo, src.hs, src.ws, False, dev)
        while lo < n:
            hi = min(n, lo + chunk)
            self.forward_chunk(src, lo, hi, mod_all, per_image, out[lo:hi], False)
            state.append((lo, hi, None))
            lo = hi
        return state

    def backward_all(self, state, g_out: Tensor, src: Source, mod_all, per_image: bool, g_in: Tensor):
        """VJP for every image: kept chunks go straight to the backward, the others recompute their forward first."""
        dev = g_out.device
        for lo, hi, saved in state:
            if saved is not None:
                self.backward_chunk(saved, g_out[lo:hi], src, lo, mod_all, per_image, g_in[lo:hi])
        pending = [(lo, hi) for lo, hi, saved in state if saved is None]
        if not pending:
            return
        lo, end = pending[0][0], pending[-1][1]
        chunk = self.chunk_size(end - lo, src.hs, src.ws, True, dev)
        scratch = torch.empty(min(chunk, end - lo), g_out.shape[1], g_out.shape[2], g_out.shape[3], device=dev,
                              dtype=torch.float32)
        while lo < end:
            hi = min(end, lo + chunk)
            saved = self.forward_chunk(src, lo, hi, mod_all, per_image, scratch[:hi - lo], True)
            self.backward_chunk(saved, g_out[lo:hi], src, lo, mod_all, per_image, g_in[lo:hi])
            del saved
            lo = hi

    # -------------------------------------------------------------------------------- whole-net kernel (1-D, one level)
    def net1d_plan(self, src: Source) -> Optional[dict]:
        """The single-launch path (csrc/net1d.hip) serves single-level 1-D nets of <= 64 channels whose convolutions are all
        k = 3, stride 1 with one padding mode -- the Lorenz score networks.  None: the per-layer / per-block kernels run."""
        if not ops.NET1D or self.depth != 1 or self.unet.spatial != 1 or src.hs != 1 or src.cctx or src.n_inner != 1:
            return None
        lev = self.levels[0]
        blocks = lev.descent + lev.ascent
        convs = [lev.head, lev.tail] + [c for b in blocks for c in (b.conv1, b.conv2)]
        if any((c.kh, c.kw, c.sh, c.sw) != (1, 3, 1, 1) or c.circular != lev.head.circular for c in convs):
            return None
        if len(blocks) > _lib_net1d_maxb() or lev.C < 2 or lev.C > 64 or src.cx > 64 or lev.tail.cout > 64:
            return None
        if any(b.act != blocks[0].act or b.ln.eps != blocks[0].ln.eps for b in blocks):
            return None
        if 64 - 2 * (2 * len(blocks) + 2) < 4:
            return None
        # the kernel forms offsets inside one image in 32 bits (net1d_check in csrc/net1d.hip: same limits, so that a view it would
        # decline takes the per-block path here instead of raising from the launch)
        if src.ws * 64 >= 1 << 30 or not _net1d_span_ok(src.sc, src.sx, src.cx, src.ws):
            return None
        return dict(lev=lev, blocks=blocks)

    def _net1d_weights(self, plan, backward: bool, cin_keep: int = 0):
        """Every convolution of the net as a [3][64][64] packing (zero padded) in the kernel's execution order, in one buffer
        (+ the biases, [conv][64], for the forward); cached until a parameter changes."""
        lev, blocks = plan['lev'], plan['blocks']
        convs = [lev.head] + [c for b in blocks for c in (b.conv1, b.conv2)] + [lev.tail]
        for cc in convs:
            cc._sync()
        key = (backward, cin_keep, tuple(cc._key for cc in convs))
        cache = self.__dict__.setdefault('_net1d_cache', {})
        hit = cache.get(backward)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        dev = lev.head.conv.weight.device
        if backward:                                     # tail^T, (conv2^T, conv1^T) of the blocks in reverse, head^T
            order = [(lev.tail, None)] + [(c, None) for b in reversed(blocks) for c in (b.conv2, b.conv1)] + [(lev.head, cin_keep)]
        else:
            order = [(cc, None) for cc in convs]
        w = torch.empty(len(order), 3 * 64 * 64, device=dev, dtype=torch.float32)
        bias = None if backward else torch.zeros(len(order), 64, device=dev, dtype=torch.float32)
        for i, (cc, keep) in enumerate(order):
            wt = cc.conv.weight.detach().contiguous()
            cout, cin = wt.shape[0], wt.shape[1]
            ops.pack_conv_weight(wt, cout, cin, 1, 3, backward, (cin if keep is None else keep) if backward else cin, w[i], 64, 64)
            if bias is not None and cc.conv.bias is not None:
                bias[i, :cout] = cc.conv.bias.detach()
        cache[backward] = (key, w, bias)
        return w, bias

    def _net1d_desc(self, plan, n: int, length: int, mod_all, lo: int, per_image: bool, backward: bool, cin_keep: int = 0):
        from ._lib import Net1dDesc
        lev, blocks = plan['lev'], plan['blocks']
        d = Net1dDesc()
        d.n, d.len, d.c, d.nblocks = n, length, lev.C, len(blocks)
        d.circular, d.unbiased = int(lev.head.circular), int(self.unbiased)
        d.act = blocks[0].act if blocks else 0
        d.eps = blocks[0].ln.eps if blocks else 1e-5
        w, bias = self._net1d_weights(plan, backward, cin_keep)
        d.w, d.bias = w.data_ptr(), None if bias is None else bias.data_ptr()
        if not backward:
            d.cin, d.cout = lev.head.cin, lev.tail.cout
        else:
            d.cin, d.cout = lev.tail.cout, cin_keep
        keep = [w, bias]
        for k, blk in enumerate(blocks):
            mod, mod_sn = self._mod_for(blk, mod_all, lo, per_image)
            d.mod[k] = None if mod is None else mod.data_ptr()
            d.mod_sn = mod_sn
            keep.append(mod)
        return d, keep

    def _net1d_forward(self, plan, src: Source, lo: int, hi: int, mod_all, per_image: bool, out: Tensor, save: bool):
        n, length = hi - lo, src.ws
        d, keep = self._net1d_desc(plan, n, length, mod_all, lo, per_image, False)
        d.x = src.x.data_ptr() + 4 * lo * src.sn_outer
        d.x_sn, d.x_sc, d.x_sx = src.sn_outer, src.sc, src.sx
        d.out = out.data_ptr()
        d.out_sn, d.out_sc, d.out_sx = out.stride(0), out.stride(1), out.stride(3)
        saved = None
        if save:
            nb, C, dev = len(plan['blocks']), plan['lev'].C, out.device
            a_s = torch.empty(max(nb, 1), n, C, length, device=dev, dtype=torch.float32)
            z_s = torch.empty_like(a_s)
            m_s = torch.empty(max(nb, 1), n, length, device=dev, dtype=torch.float32)
            r_s = torch.empty_like(m_s)
            d.a_save, d.z_save, d.save_stride = a_s.data_ptr(), z_s.data_ptr(), n * C * length
            d.mean_save, d.rstd_save, d.stat_stride = m_s.data_ptr(), r_s.data_ptr(), n * length
            saved = dict(net1d=(plan, a_s, z_s, m_s, r_s), dims=[(1, length)])
        ops.net1d_launch(d, False)
        return saved

    def _net1d_backward(self, saved, g_out: Tensor, src: Source, lo: int, mod_all, per_image: bool, g_in: Tensor):
        plan, a_s, z_s, m_s, r_s = saved['net1d']
        n, length = g_out.shape[0], src.ws
        if not _net1d_span_ok(g_out.stride(1), g_out.stride(3), g_out.shape[1], length):
            g_out = g_out.contiguous()
        d, keep = self._net1d_desc(plan, n, length, mod_all, lo, per_image, True, cin_keep=src.cx)
        d.x = g_out.data_ptr()
        d.x_sn, d.x_sc, d.x_sx = g_out.stride(0), g_out.stride(1), g_out.stride(3)
        d.out = g_in.data_ptr()
        d.out_sn, d.out_sc, d.out_sx = g_in.stride(0), g_in.stride(1), g_in.stride(3)
        C = plan['lev'].C
        d.a_save, d.z_save, d.save_stride = a_s.data_ptr(), z_s.data_ptr(), n * C * length
        d.mean_save, d.rstd_save, d.stat_stride = m_s.data_ptr(), r_s.data_ptr(), n * length
        ops.net1d_launch(d, True)

    # -------------------------------------------------------------------------------- forward
    def _mod_for(self, blk: _Block, mod_all: Optional[Tensor], lo: int, per_image: bool):
        if mod_all is None:
            return None, 0
        row = mod_all[lo:] if per_image else mod_all
        return row[:, blk.mod_off:], (self.mod_total if per_image else 0)

    def _block_fwd(self, blk: _Block, a: Tensor, mod_all, lo, per_image, saved):
        n, c, h, w = a.shape
        dev = a.device
        mod, mod_sn = self._mod_for(blk, mod_all, lo, per_image)
        c1, c2 = blk.conv1, blk.conv2
        if (c1.circular == c2.circular and (c1.sh, c1.sw, c2.sh, c2.sw) == (1, 1, 1, 1) and
                ops.block1d_eligible(c, h, c1.fwd(), c2.fwd())):
            # latency-bound 1-D nets: the whole block in one launch (block1d.hip)
            y = torch.empty_like(a)
            if saved is not None:
                mean = torch.empty(n * h * w, device=dev, dtype=torch.float32)
                rstd = torch.empty_like(mean)
                z = torch.empty_like(a)
                ops.block1d_fwd(a, mod, mod_sn, c1.fwd(), c2.fwd(), c1.circular, blk.act, blk.ln.eps, self.unbiased, y, z, mean, rstd)
                saved.append((a, mean, rstd, z))
            else:
                ops.block1d_fwd(a, mod, mod_sn, c1.fwd(), c2.fwd(), c1.circular, blk.act, blk.ln.eps, self.unbiased, y)
            return y
        mean = torch.empty(n * h * w, device=dev, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ops.ln_stats(a, mod, mod_sn, blk.ln.eps, self.unbiased, mean, rstd)
        z = torch.empty_like(a)
        pk = c1.fwd()
        z_amax = getattr(pk, 'out_amax', None)           # (f16 x 2 launches report max |z|: |act(z)| <= max(|z|, 0.28) scales conv2's input)
        d1 = launch_conv(pk, planar_source(a), z, h, w, circular=c1.circular, bias=pk.bias, mod=mod, mod_sn=mod_sn,
                         ln=(mean, rstd), out_amax=z_amax)
        if not (d1 is not None and d1.w_h2):
            z_amax = None                                # (the fp32 kernels served conv1: nothing was reported, the slot is stale)
        y = torch.empty_like(a)
        c2 = blk.conv2
        pk = c2.fwd()
        launch_conv(pk, planar_source(z), y, h, w, circular=c2.circular, bias=pk.bias, act_in=blk.act, res=a, x_amax=z_amax)
        if saved is not None:
            saved.append((a, mean, rstd, z))
        return y

    def forward_chunk(self, src: Source, lo: int, hi: int, mod_all: Optional[Tensor], per_image: bool, out: Tensor,
                      save: bool):
        """Images [lo, hi) of ``src`` -> ``out`` (n, out_channels, h, w).  Returns what the VJP needs (or None)."""
        plan = self.net1d_plan(src)
        if plan is not None:
            return self._net1d_forward(plan, src, lo, hi, mod_all, per_image, out, save)
        n = hi - lo
        dev = out.device
        L, D = self.levels, self.depth
        saved = dict(blocks={}, tails={}, dims=[]) if save else None
        skips, dims = [], []
        a, h, w = None, src.hs, src.ws
        for lvl, lev in enumerate(L):
            hd = lev.head
            pk = hd.fwd()
            if lvl == 0:
                ho, wo = conv_out_size(h, hd.kh, hd.sh), conv_out_size(w, hd.kw, hd.sw)
                a = torch.empty(n, lev.C, ho, wo, device=dev, dtype=torch.float32)
                sdesc = dict(x_ptr=src.x.data_ptr(), n=n, cx=src.cx, hs=src.hs, ws=src.ws, x_sn_outer=src.sn_outer,
                             x_sn_inner=src.sn_inner, n_inner=src.n_inner, x_n_off=lo, x_sc=src.sc, x_sy=src.sy,
                             x_sx=src.sx)
                ctx = src.ctx
                if ctx is not None and src.ctx_sn:
                    ctx = ctx[lo * src.ctx_sn:]
                launch_conv(pk, sdesc, a, ho, wo, circular=hd.circular, stride=(hd.sh, hd.sw), bias=pk.bias, ctx=ctx,
                            cctx=src.cctx, ctx_sn=src.ctx_sn)
            else:
                ho, wo = conv_out_size(h, hd.kh, hd.sh), conv_out_size(w, hd.kw, hd.sw)
                a2 = torch.empty(n, lev.C, ho, wo, device=dev, dtype=torch.float32)
                xa = None
                if ops.MULTIPLY == 'f16x2' and ops.H2_S2 and (hd.sh, hd.sw) == (2, 2) and getattr(pk, 'h2', None) is not None and a.is_contiguous():
                    xa = ops.absmax(a, pk.in_amax)       # (the f16 x 2 parity-plane form needs its input's scale: one streaming read)
                launch_conv(pk, planar_source(a), a2, ho, wo, circular=hd.circular, stride=(hd.sh, hd.sw), bias=pk.bias, x_amax=xa)
                a = a2
            h, w = ho, wo
            dims.append((h, w))
            for bi, blk in enumerate(lev.descent):
                rec = [] if save else None
                a = self._block_fwd(blk, a, mod_all, lo, per_image, rec)
                if save:
                    saved['blocks'][('d', lvl, bi)] = rec[0]
            skips.append(a)
        skips.pop()
        for lvl in reversed(range(D)):
            lev = L[lvl]
            for bi, blk in enumerate(lev.ascent):
                rec = [] if save else None
                a = self._block_fwd(blk, a, mod_all, lo, per_image, rec)
                if save:
                    saved['blocks'][('a', lvl, bi)] = rec[0]
            tl = lev.tail
            pk = tl.fwd()
            if lvl > 0:
                hu, wu = dims[lvl - 1]
                uh, uw = L[lvl].head.sh, L[lvl].head.sw
                if (h * uh, w * uw) != (hu, wu):
                    raise SdaHipError(f'U-Net level {lvl}: upsampled size {(h * uh, w * uw)} != skip size {(hu, wu)} '
                                      f'(spatial sizes must be divisible by the strides, as in the reference)')
                mean = torch.empty(n * h * w, device=dev, dtype=torch.float32)
                rstd = torch.empty_like(mean)
                ops.ln_stats(a, None, 0, lev.tail_ln.eps, self.unbiased, mean, rstd)
                t = torch.empty(n, L[lvl - 1].C, hu, wu, device=dev, dtype=torch.float32)
                launch_conv(pk, planar_source(a), t, hu, wu, circular=tl.circular, up=(uh, uw), bias=pk.bias,
                            ln=(mean, rstd), res=skips.pop())
                if save:
                    saved['tails'][lvl] = (a, mean, rstd)
                a, h, w = t, hu, wu
            else:
                launch_conv(pk, planar_source(a), out, h, w, circular=tl.circular, bias=pk.bias)
        if save:
            saved['dims'] = dims
        return saved

    # -------------------------------------------------------------------------------- backward-data
    def _block_bwd(self, blk: _Block, g: Tensor, rec, mod_all, lo, per_image, g_amax_in=None):
        """-> (gradient w.r.t. the block's input, device scalar with its max |.| or None).  g_amax_in: max |g| as its producer reported
        it (an ln_bwd launch of the previous block / tail); None: an ops.absmax pass when the f16 x 2 route needs the scale."""
        a, mean, rstd, z = rec
        n, c, h, w = a.shape
        mod, mod_sn = self._mod_for(blk, mod_all, lo, per_image)
        c1, c2 = blk.conv1, blk.conv2
        if (c1.circular == c2.circular and (c1.sh, c1.sw, c2.sh, c2.sw) == (1, 1, 1, 1) and
                ops.block1d_eligible(c, h, c1.fwd(), c2.fwd())):
            gx = torch.empty_like(a)
            ops.block1d_bwd(g, a, z, mean, rstd, mod, mod_sn, c1.bwd(), c2.bwd(), c1.circular, blk.act, self.unbiased, gx)
            return gx, None
        gz = torch.empty_like(a)
        pk2 = c2.bwd()
        g_amax = gz_amax = None
        if getattr(pk2, 'h2', None) is not None and g.is_contiguous():
            # (the producer's own report, else one streaming read of g)
            g_amax, gz_amax = (g_amax_in if g_amax_in is not None else ops.absmax(g, pk2.in_amax)), pk2.out_amax
        d2 = launch_conv(pk2, planar_source(g), gz, h, w, circular=c2.circular, dact_z=z, act_d=blk.act, x_amax=g_amax, out_amax=gz_amax)
        if not (d2 is not None and d2.w_h2):
            gz_amax = None
        gh = torch.empty_like(a)
        c1 = blk.conv1
        launch_conv(c1.bwd(), planar_source(gz), gh, h, w, circular=c1.circular, x_amax=gz_amax)
        gx = torch.empty_like(a)
        gx_amax = None
        if getattr(pk2, 'h2', None) is not None:
            # f16 x 2 route: the consumer of gx is (mostly) the previous block's conv2^T -- report max |gx| from this launch
            gx_amax = getattr(blk, '_gx_amax', None)
            if gx_amax is None or gx_amax.device != a.device:
                gx_amax = blk._gx_amax = torch.zeros(1, device=a.device, dtype=torch.float32)
        ops.ln_bwd(gh, a, h, w, mod, mod_sn, mean, rstd, self.unbiased, (1, 1), g, gx, out_amax=gx_amax)
        return gx, gx_amax

    def backward_chunk(self, saved, g_out: Tensor, src: Source, lo: int, mod_all, per_image: bool, g_in: Tensor):
        """g_out: (n, out_channels, h, w) -> g_in: (n, src.cx, hs, ws)   (context-channel gradients are not formed)."""
        if 'net1d' in saved:
            return self._net1d_backward(saved, g_out, src, lo, mod_all, per_image, g_in)
        g_out = g_out.contiguous()
        L, D = self.levels, self.depth
        dims = saved['dims']
        n = g_out.shape[0]
        dev = g_out.device
        h, w = dims[0]
        tl = L[0].tail
        g = torch.empty(n, L[0].C, h, w, device=dev, dtype=torch.float32)
        launch_conv(tl.bwd(), planar_source(g_out), g, h, w, circular=tl.circular)
        g_amax = None                                    # max |g| as reported by g's producer (ln_bwd launches), f16 x 2 route only
        g_skip = {}
        for lvl in range(D):
            lev = L[lvl]
            if lvl > 0:
                # g: gradient of (tail_lvl(a) + skip_{lvl-1}) at level lvl-1 resolution
                g_skip[lvl - 1] = g
                hu, wu = dims[lvl - 1]
                h, w = dims[lvl]
                uh, uw = lev.head.sh, lev.head.sw
                if (uh, uw) not in ((2, 2), (1, 2)):
                    raise NotImplementedError('VJP through Upsample is implemented for scale factor 2')
                tl = lev.tail
                a, mean, rstd = saved['tails'][lvl]
                # the VJP of Upsample -> conv: the transposed convolution at the fine resolution, summed over the up-sampling
                # cells -- in ONE launch where the kernel can pool in its epilogue (2 x 2: 7 of the 16 Winograd positions drop out
                # of the cell sum), else the fine-resolution gradient goes through memory and ln_bwd pools while reading it
                pooled = None
                if ops.POOLED and (uh, uw) == (2, 2) and hu == 2 * h and wu == 2 * w:
                    pooled = torch.empty(n, lev.C, h, w, device=dev, dtype=torch.float32)
                    xa = None
                    if ops.MULTIPLY == 'f16x2' and getattr(tl.bwd(), 'h2', None) is not None and g.is_contiguous():
                        xa = g_amax if g_amax is not None else ops.absmax(g, tl.bwd().in_amax)     # (the f16 x 2 parity-plane form needs g's scale)
                    if launch_conv(tl.bwd(), planar_source(g), pooled, hu, wu, circular=tl.circular, pool=(2, 2), x_amax=xa) is None:
                        pooled = None
                if pooled is None:
                    ghup = torch.empty(n, lev.C, hu, wu, device=dev, dtype=torch.float32)
                    launch_conv(tl.bwd(), planar_source(g), ghup, hu, wu, circular=tl.circular)
                g = torch.empty(n, lev.C, h, w, device=dev, dtype=torch.float32)
                g_amax = None
                if ops.MULTIPLY == 'f16x2':
                    g_amax = getattr(lev, '_g_amax', None)
                    if g_amax is None or g_amax.device != dev:
                        g_amax = lev._g_amax = torch.zeros(1, device=dev, dtype=torch.float32)
                if pooled is not None:
                    ops.ln_bwd(pooled, a, h, w, None, 0, mean, rstd, self.unbiased, (1, 1), None, g, out_amax=g_amax)
                else:
                    ops.ln_bwd(ghup, a, h, w, None, 0, mean, rstd, self.unbiased, (uh, uw), None, g, out_amax=g_amax)
            for bi in reversed(range(len(lev.ascent))):
                g, g_amax = self._block_bwd(lev.ascent[bi], g, saved['blocks'][('a', lvl, bi)], mod_all, lo, per_image, g_amax)
        for lvl in reversed(range(D)):
            lev = L[lvl]
            for bi in reversed(range(len(lev.descent))):
                g, g_amax = self._block_bwd(lev.descent[bi], g, saved['blocks'][('d', lvl, bi)], mod_all, lo, per_image, g_amax)
            hd = lev.head
            if lvl > 0:
                hu, wu = dims[lvl - 1]
                if hd.circular and ((hd.sh > 1 and hu % hd.sh) or (hd.sw > 1 and wu % hd.sw)):
                    raise NotImplementedError('VJP of a strided circular conv needs sizes divisible by the stride')
                g2 = torch.empty(n, L[lvl - 1].C, hu, wu, device=dev, dtype=torch.float32)
                skip = g_skip.pop(lvl - 1)
                classes = hd.bwd_parity() if PARITY_SPLIT and hu % hd.sh == 0 and wu % hd.sw == 0 else None
                done = False
                if (ops.MULTIPLY == 'f16x2' and ops.H2_S2 and (hd.sh, hd.sw) == (2, 2) and g.is_contiguous() and skip.is_contiguous() and
                        skip.shape == g2.shape and g.shape[2] % 16 == 0 and g.shape[3] % 16 == 0 and hd.bwd().h2_zins() is not None):
                    # f16 x 2 route: the four output parity classes as 1 / 2 / 2 / 4-tap convolutions of g on conv_h2 (PackedConv.h2_zins).
                    # (h2_only: a descriptor the kernel refuses after all launches nothing here and takes the fp32 parity-class forms below)
                    xa = g_amax if g_amax is not None else ops.absmax(g, hd.bwd().in_amax)
                    done = launch_conv(hd.bwd(), planar_source(g), g2, hu, wu, circular=hd.circular, zins=(hd.sh, hd.sw), res=skip,
                                       x_amax=xa, h2_only=True) is not None
                if not done and classes is not None and ops.PARITY4:
                    w4 = hd.bwd_parity4()
                    # (the one-launch kernel addresses the skip gradient with the OUTPUT strides: same layout required)
                    if w4 is not None and skip.is_contiguous() and skip.shape == g2.shape:
                        pk0, pad0 = classes[0][2], classes[0][3]
                        view = g2[:, :, 0::2, 0::2]
                        done = launch_conv(pk0, planar_source(g), view, view.shape[2], view.shape[3], circular=hd.circular, pad=pad0,
                                           res=skip[:, :, 0::2, 0::2], parity4_w=w4) is not None
                if done:
                    pass
                elif classes is not None:
                    gsrc = planar_source(g)
                    for py, px, pk, pad in classes:
                        ys = slice(None) if py is None else slice(py, None, 2)
                        xs = slice(None) if px is None else slice(px, None, 2)
                        view = g2[:, :, ys, xs]
                        launch_conv(pk, gsrc, view, view.shape[2], view.shape[3], circular=hd.circular, pad=pad,
                                    res=skip[:, :, ys, xs])
                else:
                    launch_conv(hd.bwd(), planar_source(g), g2, hu, wu, circular=hd.circular, zins=(hd.sh, hd.sw),
                                res=skip)
                g = g2
                g_amax = None                            # (produced by a convolution of the fp32 families: no report)
            else:
                launch_conv(hd.bwd(cin_keep=src.cx), planar_source(g), g_in, src.hs, src.ws, circular=hd.circular,
                            zins=(hd.sh, hd.sw))


# ------------------------------------------------------------------------------------------ autograd glue

class _UNetFunction(torch.autograd.Function):
    """out[n] = UNet(src[n], mod) with a hand-written VJP w.r.t. ``x`` only."""

    @staticmethod
    def forward(ctx, x: Tensor, engine: UNetEngine, src: Source, mod_all, per_image: bool, out_channels: int):
        need = ctx.needs_input_grad[0]
        dev = x.device
        n = src.n
        # output spatial size of level 0
        hd = engine.levels[0].head
        ho, wo = conv_out_size(src.hs, hd.kh, hd.sh), conv_out_size(src.ws, hd.kw, hd.sw)
        # a channel-last trajectory (MCScoreWrapper's transposed view of a (B, L, C) tensor) through the whole-net 1-D kernel:
        # the output is laid out channel-last too, so that the wrapper's transpose back is a contiguous tensor -- the kernel
        # writes through strides, no copy kernel on either side (and likewise for the cotangent / input gradient below)
        ctx.channel_last = bool(src.hs == 1 and src.sc == 1 and src.sx == src.cx and src.cx > 1 and engine.net1d_plan(src) is not None)
        if ctx.channel_last:
            out = torch.empty(n, ho, wo, out_channels, device=dev, dtype=torch.float32).permute(0, 3, 1, 2)
        else:
            out = torch.empty(n, out_channels, ho, wo, device=dev, dtype=torch.float32)
        ctx.engine, ctx.src, ctx.mod_all, ctx.per_image = engine, src, mod_all, per_image
        ctx.x_shape = x.shape
        ctx.vjp_state = engine.forward_all(src, mod_all, per_image, out, need)
        return out

    @staticmethod
    def backward(ctx, g_out: Tensor):
        engine, src = ctx.engine, ctx.src
        if ctx.channel_last:                             # (strided cotangent in, channel-last gradient out: see forward)
            g_in = torch.empty(src.n, src.hs, src.ws, src.cx, device=g_out.device, dtype=torch.float32).permute(0, 3, 1, 2)
        else:
            g_out = g_out.contiguous()
            g_in = torch.empty(src.n, src.cx, src.hs, src.ws, device=g_out.device, dtype=torch.float32)
        engine.backward_all(ctx.vjp_state, g_out, src, ctx.mod_all, ctx.per_image, g_in)
        return g_in.reshape(ctx.x_shape), None, None, None, None, None


def run_unet(unet, src: Source, emb: Tensor, grad_anchor: Tensor) -> Tensor:
    """Common entry: ``emb`` (T, mod_features) with T in {1, n}; returns (n, out_channels, h, w)."""
    global _warned_training
    engine = unet.engine()
    T = emb.shape[0]
    if T not in (1, src.n):
        raise SdaHipError(f'time embedding batch {T} does not broadcast against {src.n} images')
    per_image = T != 1
    mod_all = engine.modulation(emb) if engine.mod_total > 0 else None
    if torch.is_grad_enabled() and not grad_anchor.requires_grad and not _warned_training:
        if any(p.requires_grad for p in unet.parameters()):
            _warned_training = True
            import warnings
            warnings.warn('sda_amd implements the sampling hot path: gradients flow to the network INPUT only; '
                          'parameter gradients (training) are out of scope and are not computed.')
    return _UNetFunction.apply(grad_anchor, engine, src, mod_all, per_image, unet.out_channels)


def source_from_tensor(x: Tensor, spatial: int):
    """(…, C, *spatial) strided tensor -> (Source fields, n).  Collapses the batch dims without copying if possible."""
    dims = spatial + 1
    batch = x.shape[:-dims]
    n = 1
    for b in batch:
        n *= b
    # try to express the batch dims with a single stride
    xv = x
    if len(batch) != 1:
        try:
            xv = x.view(n, *x.shape[-dims:])
        except RuntimeError:
            xv = x.reshape(n, *x.shape[-dims:])     # copies only when the batch dims are not collapsible
    c = xv.shape[1]
    if spatial == 1:
        hs, ws = 1, xv.shape[2]
        sy, sx = 0, xv.stride(2)
    else:
        hs, ws = xv.shape[2], xv.shape[3]
        sy, sx = xv.stride(2), xv.stride(3)
    return xv, Source(x=xv, n=n, cx=c, hs=hs, ws=ws, sn_outer=xv.stride(0) if n > 1 else 0, sc=xv.stride(1), sy=sy, sx=sx)


def attach_context(src: Source, c: Optional[Tensor], spatial: int):
    """Context channels (score.py:87): broadcast over the batch when possible, else one block per image."""
    if c is None:
        return
    dims = spatial + 1
    cc = c.shape[-dims]
    per = 1
    for s in c.shape[-dims:]:
        per *= s
    if tuple(c.shape[-spatial:]) != ((src.ws,) if spatial == 1 else (src.hs, src.ws)):
        raise SdaHipError('context spatial shape does not match x')
    if c.numel() == per:
        src.ctx, src.cctx, src.ctx_sn = c.reshape(-1).contiguous(), cc, 0
    else:
        full = c.expand(*src.x.shape[:1], *c.shape[-dims:]) if c.dim() == dims + 1 else None
        if full is None or full.shape[0] != src.n:
            raise SdaHipError('context batch shape must be broadcast-free or match the flattened batch of x')
        src.ctx, src.cctx, src.ctx_sn = full.contiguous().reshape(-1), cc, per


def unet_apply(unet, x: Tensor, y: Tensor) -> Tensor:
    """``UNet.forward(x, y)`` (nn.py:184-206): x (N, C, *spatial), y (N|1, mod_features)."""
    ops._dev(x, y)
    if unet.spatial == 3:
        from .engine3d import run_unet3d
        out = run_unet3d(unet, x.reshape((-1,) + tuple(x.shape[-4:])), y.reshape(-1, y.shape[-1]))
        return out.reshape(tuple(x.shape[:-4]) + tuple(out.shape[1:]))
    xv, src = source_from_tensor(x, unet.spatial)
    out = run_unet(unet, src, y.reshape(-1, y.shape[-1]), x)
    return out if unet.spatial == 2 else out[:, :, 0]


def block_forward_standalone(block, x: Tensor, y: Tensor) -> Tensor:
    """``ModResidualBlock.forward`` outside a U-Net (nn.py:27-28): same three kernels as inside the engine."""
    ops._dev(x, y)
    spatial = x.dim() - 2
    if spatial == 3:
        from .engine3d import block_forward_standalone3d
        return block_forward_standalone3d(block, x, y)
    if spatial not in (1, 2):
        raise NotImplementedError
    xs = x.contiguous()
    a = xs if spatial == 2 else xs.unsqueeze(2)
    n, c = a.shape[:2]
    eng = UNetEngine.__new__(UNetEngine)
    from .nn import LN_UNBIASED
    eng.unbiased, eng.mod_total = LN_UNBIASED, c
    blk = _Block(block, c, 0)
    lin = blk.project
    mod = ops.linear_small(y.reshape(-1, y.shape[-1]).contiguous(), lin.weight.detach().contiguous(),
                           lin.bias.detach().contiguous())
    if mod.shape[0] not in (1, n):
        raise SdaHipError('modulation batch does not broadcast')
    out = eng._block_fwd(blk, a.contiguous(), mod, 0, mod.shape[0] != 1, None)
    return out if spatial == 2 else out[:, :, 0]
