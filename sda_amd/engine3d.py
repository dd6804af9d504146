"""``UNet(spatial=3)``: the reference builds the same U-Net around ``nn.Conv3d`` (sda/nn.py:114-118, 148-206).

No reference experiment instantiates it, so this is the path's general member: every convolution of the forward pass and of
the input VJP is one launch of ``sda_conv3d`` (csrc/conv3d.hip: implicit GEMM on the fp32 matrix cores, stride / nearest
up-sampling / zero insertion / circular wrap in its index maps, bias / activation / act' / residual in its epilogue); LayerNorm
runs on the planar kernels the 2-D path uses for its statistics and backward (``sda_ln_stats / _apply / _bwd``: the three spatial
axes are one plane to them).  Activations are planar ``(N, C, D, H, W)``; everything a VJP needs is kept (no chunking).

    block   (nn.py:27-28, 131-142):  z = conv1(LN(a + m));  y = a + conv2(act(z))            VJP: conv2^T x act'(z), conv1^T, LN'
    head_l  (nn.py:152-159):         stride-s convolution                                      VJP: zero-inserted transposed conv + skip
    tail_l  (nn.py:161-169):         conv(up_s(LN(a))) + skip                                  VJP: conv^T, s^3-cell sums, LN'
"""
from typing import List, Optional

import torch
from torch import Tensor

from . import _lib, ops
from ._lib import SdaHipError


class _Conv3d:
    """Packed forward / input-VJP weights of one ``nn.Conv3d`` (re-packed when the parameter changes)."""

    MAX_IMAGES = 65535

    def __init__(self, conv):
        if tuple(conv.dilation) != (1, 1, 1) or conv.groups != 1:
            raise NotImplementedError('dilated / grouped convolutions have no gfx950 kernel')
        if conv.padding_mode not in ('zeros', 'circular'):
            raise NotImplementedError(f"padding_mode={conv.padding_mode!r} (supported: 'zeros', 'circular')")
        if any(k % 2 == 0 for k in conv.kernel_size):
            raise NotImplementedError('even kernel sizes are not supported')
        self.conv = conv
        self.k = tuple(conv.kernel_size)
        self.stride = tuple(conv.stride)
        self.pad = tuple(k // 2 for k in self.k)
        self.circular = conv.padding_mode == 'circular'
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self._key, self._packs = None, {}

    def invalidate(self):
        self._key = None

    def pack(self, transpose: bool) -> Tensor:
        w = self.conv.weight
        key = (w.data_ptr(), ops.tensor_version(w), str(w.device))
        if key != self._key:
            self._key, self._packs = key, {}
        if transpose not in self._packs:
            lib = _lib.load()
            nfl = lib.sda_conv3d_packed_floats(self.cout, self.cin, *self.k, int(transpose))
            dst = torch.empty(nfl, device=w.device, dtype=torch.float32)
            wc = w.detach().contiguous()
            ops._dev(wc)
            _lib.check(lib.sda_pack_conv3d_weight(wc.data_ptr(), self.cout, self.cin, *self.k, int(transpose), dst.data_ptr(),
                                                  ops._stream()), 'sda_pack_conv3d_weight')
            self._packs[transpose] = dst
        return self._packs[transpose]

    def bias(self) -> Optional[Tensor]:
        b = self.conv.bias
        return None if b is None else b.detach()

    def out_size(self, size):
        return tuple((n + 2 * p - k) // s + 1 for n, p, k, s in zip(size, self.pad, self.k, self.stride))

    # ---- launches -------------------------------------------------------------------------------------------------
    def _launch(self, x: Tensor, w: Tensor, out: Tensor, cin, cout, *, pad, stride=(1, 1, 1), up=(1, 1, 1), dil=(1, 1, 1),
                bias=None, act=0, act_in=0, z=None, res=None):
        ops._dev(x, w, out, bias, z, res)
        for t in (x, out, z, res):
            if t is not None and not t.is_contiguous():
                raise SdaHipError('conv3d operands are planar and contiguous')
        if x.shape[1] != cin or out.shape[1] != cout or out.shape[0] != x.shape[0]:
            raise SdaHipError(f'conv3d: operand shapes {tuple(x.shape)} -> {tuple(out.shape)} do not match the {cin} -> {cout} operator')
        d = _lib.Conv3dDesc()
        d.w, d.bias, d.cin, d.cout = w.data_ptr(), ops._ptr(bias), cin, cout
        for a in range(3):
            d.in_size[a], d.out_size[a] = x.shape[2 + a], out.shape[2 + a]
            d.k[a], d.pad[a], d.stride[a], d.up[a], d.dil[a] = self.k[a], pad[a], stride[a], up[a], dil[a]
        d.circular, d.act, d.act_in = int(self.circular), act, act_in
        lib, n = _lib.load(), x.shape[0]
        for lo in range(0, n, self.MAX_IMAGES):              # (the image index is the grid's z dimension: 65 535 per launch)
            hi = min(n, lo + self.MAX_IMAGES)
            d.n = hi - lo
            d.x, d.out = x[lo:hi].data_ptr(), out[lo:hi].data_ptr()
            d.z, d.res = (None if z is None else z[lo:hi].data_ptr()), (None if res is None else res[lo:hi].data_ptr())
            _lib.check(lib.sda_conv3d(d, ops._stream()), 'sda_conv3d')
        return out

    def forward(self, x: Tensor, *, up=(1, 1, 1), act_in=0, res=None) -> Tensor:
        size = tuple(s * u for s, u in zip(x.shape[2:], up))
        out = torch.empty((x.shape[0], self.cout) + self.out_size(size), device=x.device, dtype=torch.float32)
        return self._launch(x, self.pack(False), out, self.cin, self.cout, pad=self.pad, stride=self.stride, up=up,
                            bias=self.bias(), act_in=act_in, res=res)

    def vjp(self, g: Tensor, in_size, *, act=0, z=None, res=None) -> Tensor:
        """Gradient w.r.t. the convolution's (possibly up-sampled) input of extent ``in_size``: the transposed convolution --
        flipped taps, zero insertion for a strided layer --, optionally x act'(z), optionally + res."""
        if self.circular and any(n != go * s for n, go, s in zip(in_size, g.shape[2:], self.stride)):
            # (the kernel wraps the zero-inserted gradient with period out x stride; only then is that the forward's period)
            raise SdaHipError(f'circular strided Conv3d VJP: input extent {tuple(in_size)} is not output {tuple(g.shape[2:])} x stride '
                              f'{tuple(self.stride)}')
        out = torch.empty((g.shape[0], self.cin) + tuple(in_size), device=g.device, dtype=torch.float32)
        pad = tuple(k - 1 - p for k, p in zip(self.k, self.pad))
        return self._launch(g, self.pack(True), out, self.cout, self.cin, pad=pad, dil=self.stride, act=act, z=z, res=res)


class _Block3d:
    def __init__(self, block, width: int, mod_off: int):
        from .nn import activation_id
        residue = block.residue
        self.ln = residue[0]
        self.conv1, self.act, self.conv2 = _Conv3d(residue[1]), activation_id(residue[2]), _Conv3d(residue[3])
        self.project = block.project[0]
        self.width, self.mod_off = width, mod_off


def _plane(t: Tensor) -> Tensor:
    """(N, C, D, H, W) -> the (N, C, D*H*W) view the planar LayerNorm kernels take."""
    return t.view(t.shape[0], t.shape[1], -1)


def _pool_sum(g: Tensor, f) -> Tensor:
    n, c, d, h, w = g.shape
    out = torch.empty(n, c, d // f[0], h // f[1], w // f[2], device=g.device, dtype=torch.float32)
    _lib.check(_lib.load().sda_pool3d_sum(g.data_ptr(), n * c, out.shape[2], out.shape[3], out.shape[4], f[0], f[1], f[2],
                                          out.data_ptr(), ops._stream()), 'sda_pool3d_sum')
    return out


class UNet3dEngine:
    def __init__(self, unet):
        from .nn import LN_UNBIASED
        self.unet, self.unbiased = unet, LN_UNBIASED
        self.depth = D = len(unet.hidden_blocks)
        self.heads, self.tails, self.tail_ln, self.descent, self.ascent = [], [], [], [], []
        off = 0
        for lvl in range(D):
            C = unet.hidden_channels[lvl]
            j = D - 1 - lvl                                   # tails / ascent are stored deepest first (nn.py:179-182)
            self.heads.append(_Conv3d(unet.heads[lvl] if lvl == 0 else unet.heads[lvl][0]))
            self.tails.append(_Conv3d(unet.tails[j] if lvl == 0 else unet.tails[j][2]))
            self.tail_ln.append(None if lvl == 0 else unet.tails[j][0])
            blocks = []
            for blk in unet.descent[lvl]:
                blocks.append(_Block3d(blk, C, off)); off += C
            self.descent.append(blocks)
            blocks = []
            for blk in unet.ascent[j]:
                blocks.append(_Block3d(blk, C, off)); off += C
            self.ascent.append(blocks)
        self.mod_total = off
        self.scale = tuple(unet.stride)
        self._proj_key = self._proj = None

    def _blocks(self) -> List[_Block3d]:
        return [b for lvl in range(self.depth) for b in self.descent[lvl] + self.ascent[lvl]]

    def invalidate(self):
        for c in self.heads + self.tails:
            c.invalidate()
        for b in self._blocks():
            b.conv1.invalidate()
            b.conv2.invalidate()
        self._proj_key = None

    def modulation(self, emb: Tensor) -> Optional[Tensor]:
        """emb (T, mod_features) -> every block's modulation vector, (T, mod_total), in one small launch."""
        blocks = self._blocks()
        if not blocks:
            return None
        key = tuple((b.project.weight.data_ptr(), ops.tensor_version(b.project.weight), ops.tensor_version(b.project.bias)) for b in blocks)
        if key != self._proj_key:
            w = torch.cat([b.project.weight.detach() for b in blocks], dim=0).contiguous()
            bias = torch.cat([b.project.bias.detach() for b in blocks], dim=0).contiguous()
            self._proj_key, self._proj = key, (w, bias)
        return ops.linear_small(emb.contiguous(), *self._proj)

    def bytes_per_image(self, *a, **k):
        return None

    # ---------------------------------------------------------------------------------------------------- forward
    def _mod(self, blk: _Block3d, mod_all: Optional[Tensor], per_image: bool):
        if mod_all is None:
            return None, 0
        return mod_all[:, blk.mod_off:], (self.mod_total if per_image else 0)

    def _ln(self, a: Tensor, mod, mod_sn, eps: float):
        n = a.shape[0]
        mean = torch.empty(n * a[0, 0].numel(), device=a.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ops.ln_stats(_plane(a), mod, mod_sn, eps, self.unbiased, mean, rstd)
        y = torch.empty_like(a)
        ops.ln_apply(_plane(a), mod, mod_sn, mean, rstd, _plane(y))
        return y, mean, rstd

    def _block_fwd(self, blk: _Block3d, a: Tensor, mod_all, per_image, saved):
        mod, mod_sn = self._mod(blk, mod_all, per_image)
        xn, mean, rstd = self._ln(a, mod, mod_sn, blk.ln.eps)
        z = blk.conv1.forward(xn)
        y = blk.conv2.forward(z, act_in=blk.act, res=a)
        if saved is not None:
            saved.append((a, mean, rstd, z))
        return y

    def forward_all(self, x: Tensor, mod_all, per_image: bool, save: bool):
        """x (N, in_channels, D, H, W) contiguous -> (out, what the VJP needs | None)."""
        D = self.depth
        saved = dict(blocks={}, tails={}, sizes=[]) if save else None
        skips = []
        a = x
        for lvl in range(D):
            a = self.heads[lvl].forward(a)
            if save:
                saved['sizes'].append(tuple(a.shape[2:]))
            for i, blk in enumerate(self.descent[lvl]):
                rec = [] if save else None
                a = self._block_fwd(blk, a, mod_all, per_image, rec)
                if save:
                    saved['blocks'][('d', lvl, i)] = rec[0]
            skips.append(a)
        skips.pop()
        for lvl in range(D - 1, -1, -1):
            for i, blk in enumerate(self.ascent[lvl]):
                rec = [] if save else None
                a = self._block_fwd(blk, a, mod_all, per_image, rec)
                if save:
                    saved['blocks'][('a', lvl, i)] = rec[0]
            if lvl > 0:
                skip = skips.pop()
                if tuple(s * f for s, f in zip(a.shape[2:], self.scale)) != tuple(skip.shape[2:]):
                    raise SdaHipError(f'U-Net level {lvl}: {tuple(a.shape[2:])} x {self.scale} does not match the skip '
                                      f'{tuple(skip.shape[2:])} (spatial sizes must be divisible by the strides, as in the reference)')
                xn, mean, rstd = self._ln(a, None, 0, self.tail_ln[lvl].eps)
                if save:
                    saved['tails'][lvl] = (a, mean, rstd)
                a = self.tails[lvl].forward(xn, up=self.scale, res=skip)
            else:
                a = self.tails[0].forward(a)
        return a, saved

    # ---------------------------------------------------------------------------------------------------- input VJP
    def _block_bwd(self, blk: _Block3d, g: Tensor, rec, mod_all, per_image):
        a, mean, rstd, z = rec
        mod, mod_sn = self._mod(blk, mod_all, per_image)
        size = tuple(a.shape[2:])
        gz = blk.conv2.vjp(g, size, act=blk.act, z=z)
        gh = blk.conv1.vjp(gz, size)
        gx = torch.empty_like(a)
        ops.ln_bwd(_plane(gh), _plane(a), 1, a[0, 0].numel(), mod, mod_sn, mean, rstd, self.unbiased, (1, 1), _plane(g), _plane(gx))
        return gx

    def backward_all(self, saved, g_out: Tensor, in_size, mod_all, per_image: bool) -> Tensor:
        D = self.depth
        sizes = saved['sizes']
        g = self.tails[0].vjp(g_out.contiguous(), sizes[0])
        g_skip = {}
        for lvl in range(D):
            if lvl > 0:
                g_skip[lvl - 1] = g
                a, mean, rstd = saved['tails'][lvl]
                fine = self.tails[lvl].vjp(g, sizes[lvl - 1])             # gradient of the up-sampled, normalised tensor
                pooled = _pool_sum(fine, self.scale)
                ga = torch.empty_like(a)
                ops.ln_bwd(_plane(pooled), _plane(a), 1, a[0, 0].numel(), None, 0, mean, rstd, self.unbiased, (1, 1), None, _plane(ga))
                g = ga
            for i in range(len(self.ascent[lvl]) - 1, -1, -1):
                g = self._block_bwd(self.ascent[lvl][i], g, saved['blocks'][('a', lvl, i)], mod_all, per_image)
        for lvl in range(D - 1, -1, -1):
            for i in range(len(self.descent[lvl]) - 1, -1, -1):
                g = self._block_bwd(self.descent[lvl][i], g, saved['blocks'][('d', lvl, i)], mod_all, per_image)
            if lvl > 0:
                g = self.heads[lvl].vjp(g, sizes[lvl - 1], res=g_skip[lvl - 1])
        return self.heads[0].vjp(g, in_size)


class _UNet3dFunction(torch.autograd.Function):
    """out = UNet3d(x, mod) with the hand-written VJP w.r.t. ``x`` (parameter gradients are never formed)."""

    @staticmethod
    def forward(ctx, x: Tensor, engine: UNet3dEngine, mod_all, per_image: bool):
        need = ctx.needs_input_grad[0]
        out, saved = engine.forward_all(x.detach().contiguous(), mod_all, per_image, need)
        ctx.engine, ctx.saved, ctx.mod_all, ctx.per_image, ctx.in_size = engine, saved, mod_all, per_image, tuple(x.shape[2:])
        return out

    @staticmethod
    def backward(ctx, g_out: Tensor):
        g = ctx.engine.backward_all(ctx.saved, g_out, ctx.in_size, ctx.mod_all, ctx.per_image)
        return g, None, None, None


def run_unet3d(unet, x: Tensor, emb: Tensor) -> Tensor:
    """x (N, in_channels, D, H, W), emb (T, mod_features) with T in {1, N} -> (N, out_channels, D, H, W)."""
    ops._dev(x, emb)
    engine = unet.engine()
    T = emb.shape[0]
    if T not in (1, x.shape[0]):
        raise SdaHipError(f'time embedding batch {T} does not broadcast against {x.shape[0]} images')
    mod_all = engine.modulation(emb)
    return _UNet3dFunction.apply(x, engine, mod_all, T != 1)


def block_forward_standalone3d(block, x: Tensor, y: Tensor) -> Tensor:
    """``ModResidualBlock.forward`` on a 5-D tensor outside a U-Net (nn.py:27-28): the block's three launches."""
    from .nn import LN_UNBIASED
    xs = x.contiguous()
    n, c = xs.shape[:2]
    eng = UNet3dEngine.__new__(UNet3dEngine)
    eng.unbiased, eng.mod_total = LN_UNBIASED, c
    blk = _Block3d(block, c, 0)
    lin = blk.project
    mod = ops.linear_small(y.reshape(-1, y.shape[-1]).contiguous(), lin.weight.detach().contiguous(), lin.bias.detach().contiguous())
    if mod.shape[0] not in (1, n):
        raise SdaHipError('modulation batch does not broadcast')
    return eng._block_fwd(blk, xs, mod, mod.shape[0] != 1, None)
