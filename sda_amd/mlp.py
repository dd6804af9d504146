"""ResMLP / ScoreNet path (the Lorenz *local* kernel; sda/nn.py:31-71, sda/score.py:38-63) on MI355X.

Row-major ``(rows, features)`` activations; kernels in csrc/linear.hip:
  * sda_linear            : y = act(x W^T + b) (* act'(z)) (+ res) on the fp32 matrix cores; also gx = gy W
  * sda_row_ln / _bwd     : zuko LayerNorm over the last axis, one 64-lane wavefront per row (shuffle reductions)

Per residual block ``x + Lin2(act(Lin1(LN(x))))`` the forward is three launches (row_ln, linear, linear with the
activation applied to its input and the residual in its epilogue) and saves ``x``, the LN statistics and the
pre-activation ``z``; the input gradient is three launches again (gy W2 * act'(z), gz W1, LN backward + g).
"""
import os
from typing import List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib, ops


def _rows(x: Tensor):
    xs = x.contiguous()
    return xs.reshape(-1, xs.shape[-1])


def _is_res_block(block) -> bool:
    from .nn import LayerNorm
    return (isinstance(block, nn.Sequential) and len(block) == 4 and isinstance(block[0], LayerNorm)
            and isinstance(block[1], nn.Linear) and isinstance(block[3], nn.Linear) and block[0].dim == -1)


# ---------------------------------------------------------------------------------------------- whole-MLP kernels (csrc/mlp1d.hip)
FUSED = os.environ.get('SDA_MLP_FUSED', '1') != '0'
_MLP_W = 128


def _mf(out_f: int) -> int:
    """D fragments (16 features each) of an output width: padded to 16 or 128 features."""
    return 1 if out_f <= 16 else 8


def _kq(in_f: int) -> int:
    """K quads (16 values each) of a contraction length: padded to 16 / 64 / 128."""
    return 1 if in_f <= 16 else (4 if in_f <= 64 else 8)


_PIECE = 4096          # floats per staging piece of the kernel (slabs are zero padded to whole pieces in memory)


def _slab(W: Tensor) -> Tensor:
    """The LDS slab of one GEMM y = W x, W [out][in] (csrc/mlp1d.hip): [fragment m][k quad sq][lane = 16 kq + li][4] with element e of lane
    (kq, li) = Wp[16 m + li][16 sq + 4 kq + e], Wp zero padded; padded with zeros to whole staging pieces."""
    o, i = W.shape
    mf, kq = _mf(o), _kq(i)
    Wp = torch.zeros(16 * mf, 16 * kq, device=W.device, dtype=torch.float32)
    Wp[:o, :i] = W
    mat = Wp.view(mf, 16, kq, 4, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)
    return torch.nn.functional.pad(mat, (0, -mat.numel() % _PIECE))


class _FusedPlan:
    """The GEMM list of a ResMLP for sda_mlp_fwd / sda_mlp_bwd and its packed (zero-padded) weights, forward and transposed; rebuilt when
    a parameter changes (pointer / version keys, as the convolution caches)."""

    def __init__(self, layers):
        from .nn import LN_UNBIASED, activation_id
        self.gemms = []                 # (kind, in_f, out_f, Linear)
        acts, epss = set(), set()
        for layer in layers:
            if isinstance(layer, nn.Linear):
                self.gemms.append((0, layer.in_features, layer.out_features, layer))
            else:
                ln, l1, act_m, l2 = layer[0], layer[1], layer[2], layer[3]
                acts.add(activation_id(act_m))
                epss.add(float(ln.eps))
                self.gemms.append((1, l1.in_features, l1.out_features, l1))
                self.gemms.append((2, l2.in_features, l2.out_features, l2))
        self.ok = (1 <= len(self.gemms) <= _lib.MLP_MAXG and len(acts) <= 1 and len(epss) <= 1 and
                   all(i <= _MLP_W and o <= _MLP_W and lin.bias is not None for _, i, o, lin in self.gemms) and
                   all(i == o for k, i, o, _ in self.gemms if k) and
                   all(i >= 2 for k, i, _o, _ in self.gemms if k == 1 and LN_UNBIASED) and   # (unbiased LayerNorm of one feature: C ABI says UNSUPPORTED)
                   all(self.gemms[j][1] == self.gemms[j - 1][2] for j in range(1, len(self.gemms))))
        self.act = acts.pop() if acts else 0
        self.eps = epss.pop() if epss else 1e-5
        self.unbiased = LN_UNBIASED
        self.nres = sum(1 for k, *_ in self.gemms if k == 2)
        self._key = None

    def _pack(self):
        key = tuple((lin.weight.data_ptr(), ops.tensor_version(lin.weight), ops.tensor_version(lin.bias), str(lin.weight.device)) for *_, lin in self.gemms)
        if key == self._key:
            return
        fw, bw, bs, w_off, b_off, wn, bn = [], [], [], [], [], 0, 0
        for _k, i, o, lin in self.gemms:
            W = lin.weight.detach().to(torch.float32)
            sf, sb = _slab(W), _slab(W.t())
            # (forward and transposed slabs share one offsets table: laid out at the larger of the two sizes)
            size = max(sf.numel(), sb.numel())
            fw.append(torch.nn.functional.pad(sf, (0, size - sf.numel())))
            bw.append(torch.nn.functional.pad(sb, (0, size - sb.numel())))
            b = torch.zeros(16 * _mf(o), device=W.device, dtype=torch.float32)
            b[:o] = lin.bias.detach()
            bs.append(b)
            w_off.append(wn); b_off.append(bn)
            wn += size
            bn += b.numel()
        self.wf, self.wb, self.bias, self.w_off, self.b_off, self._key = torch.cat(fw), torch.cat(bw), torch.cat(bs), w_off, b_off, key

    def desc(self, rows: int, backward: bool):
        self._pack()
        d = _lib.MlpDesc()
        d.rows, d.ngemm, d.act, d.unbiased, d.eps = rows, len(self.gemms), self.act, int(self.unbiased), self.eps
        for g, (k, i, o, _lin) in enumerate(self.gemms):
            d.kind[g], d.in_f[g], d.out_f[g], d.w_off[g], d.b_off[g] = k, i, o, self.w_off[g], self.b_off[g]
        d.w = (self.wb if backward else self.wf).data_ptr()
        d.bias = self.bias.data_ptr()
        return d


def _fused_plan(layers) -> Optional['_FusedPlan']:
    """The cached plan of this layer list (kept on its first module), or None when the whole-MLP kernels do not take it."""
    if not FUSED or not layers:
        return None
    holder = layers[0]
    key = tuple(id(l) for l in layers)
    hit = holder.__dict__.get('_sda_mlp_plan')
    if hit is None or hit[0] != key:
        hit = (key, _FusedPlan(layers))
        holder.__dict__['_sda_mlp_plan'] = hit
    return hit[1] if hit[1].ok else None


class _FusedMLPFunction(torch.autograd.Function):
    """The whole layer chain in one launch (sda_mlp_fwd) and its input VJP in one more (sda_mlp_bwd)."""

    @staticmethod
    def forward(ctx, x: Tensor, plan: _FusedPlan):
        need = ctx.needs_input_grad[0]
        rows = x.shape[0]
        d = plan.desc(rows, False)
        out = torch.empty(rows, plan.gemms[-1][2], device=x.device, dtype=torch.float32)
        d.x, d.x_ld, d.out, d.out_ld = x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0)
        saves = None
        if need and plan.nres:
            a_s = torch.empty(plan.nres, rows, _MLP_W, device=x.device, dtype=torch.float32)
            z_s = torch.empty_like(a_s)
            m_s = torch.empty(plan.nres, rows, device=x.device, dtype=torch.float32)
            r_s = torch.empty_like(m_s)
            d.a_save, d.z_save, d.save_stride, d.save_ld = a_s.data_ptr(), z_s.data_ptr(), rows * _MLP_W, _MLP_W
            d.mean_save, d.rstd_save, d.stat_stride = m_s.data_ptr(), r_s.data_ptr(), rows
            saves = (a_s, z_s, m_s, r_s)
        ops.mlp_launch(d, False)
        ctx.plan, ctx.saves, ctx.rows = plan, saves, rows
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        plan, rows = ctx.plan, ctx.rows
        g = g.contiguous()
        d = plan.desc(rows, True)
        gx = torch.empty(rows, plan.gemms[0][1], device=g.device, dtype=torch.float32)
        d.x, d.x_ld, d.out, d.out_ld = g.data_ptr(), g.stride(0), gx.data_ptr(), gx.stride(0)
        if ctx.saves is not None:
            a_s, z_s, m_s, r_s = ctx.saves
            d.a_save, d.z_save, d.save_stride, d.save_ld = a_s.data_ptr(), z_s.data_ptr(), rows * _MLP_W, _MLP_W
            d.mean_save, d.rstd_save, d.stat_stride = m_s.data_ptr(), r_s.data_ptr(), rows
        ops.mlp_launch(d, True)
        return gx, None


class _MLPFunction(torch.autograd.Function):
    """A chain of ``nn.Linear`` and residual blocks on (rows, features); VJP w.r.t. the input only."""

    @staticmethod
    def forward(ctx, x: Tensor, layers):
        from .nn import LN_UNBIASED, activation_id
        need = ctx.needs_input_grad[0]
        saved: List[tuple] = []
        h = x
        for layer in layers:
            if isinstance(layer, nn.Linear):
                bias = None if layer.bias is None else layer.bias.detach()
                h2 = ops.linear(h, layer.weight.detach(), bias)
                saved.append(('lin', layer))
                h = h2
            else:
                ln, l1, act_m, l2 = layer[0], layer[1], layer[2], layer[3]
                act = activation_id(act_m)
                rows = h.shape[0]
                mean = torch.empty(rows, device=h.device, dtype=torch.float32)
                rstd = torch.empty_like(mean)
                hn = torch.empty_like(h)
                ops.row_ln(h, ln.eps, LN_UNBIASED, hn, mean, rstd)
                z = ops.linear(hn, l1.weight.detach(), None if l1.bias is None else l1.bias.detach())
                out = ops.linear(z, l2.weight.detach(), None if l2.bias is None else l2.bias.detach(), act_in=act, res=h)
                saved.append(('res', layer, h if need else None, mean, rstd, z if need else None, act))
                h = out
        ctx.saved_chain = saved if need else None
        return h

    @staticmethod
    def backward(ctx, g: Tensor):
        from .nn import LN_UNBIASED
        g = g.contiguous()
        for rec in reversed(ctx.saved_chain):
            if rec[0] == 'lin':
                g = ops.linear(g, rec[1].weight.detach(), None, trans_w=True)
            else:
                _, layer, x, mean, rstd, z, act = rec
                l1, l2 = layer[1], layer[3]
                gz = ops.linear(g, l2.weight.detach(), None, trans_w=True, dact_z=z, act_d=act)
                gh = ops.linear(gz, l1.weight.detach(), None, trans_w=True)
                gx = torch.empty_like(x)
                ops.row_ln_bwd(gh, x, mean, rstd, LN_UNBIASED, g, gx)
                g = gx
        return g, None


def _run(layers, x: Tensor) -> Tensor:
    ops._dev(x)
    xr = _rows(x)
    layers = list(layers)
    plan = _fused_plan(layers)
    if plan is not None and xr.is_cuda and xr.shape[0] > 0 and xr.stride(1) == 1 and xr.shape[1] == plan.gemms[0][1]:
        out = _FusedMLPFunction.apply(xr, plan)
        return out.reshape(*x.shape[:-1], out.shape[-1])
    out = _MLPFunction.apply(xr, layers)
    return out.reshape(*x.shape[:-1], out.shape[-1])


def row_layer_norm(x: Tensor, eps: float, unbiased: bool) -> Tensor:
    ops._dev(x)
    xr = _rows(x)
    y = torch.empty_like(xr)
    ops.row_ln(xr, eps, unbiased, y)
    return y.reshape(x.shape)


def try_fused_residual_mlp(block, x: Tensor):
    """ResidualBlock(LayerNorm(), Linear, act, Linear) -> fused path; anything else -> None (caller falls back)."""
    if not _is_res_block(block):
        return None
    return _run([block], x)


def resmlp_forward(mlp, x: Tensor) -> Tensor:
    layers = list(mlp)
    if not all(isinstance(l, nn.Linear) or _is_res_block(l) for l in layers):
        raise NotImplementedError('ResMLP with custom layers has no gfx950 path')
    return _run(layers, x)
