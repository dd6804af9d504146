"""ResMLP / ScoreNet path (the Lorenz *local* kernel; sda/nn.py:31-71, sda/score.py:38-63) on MI355X.

Row-major ``(rows, features)`` activations; kernels in csrc/linear.hip:
  * sda_linear            : y = act(x W^T + b) (* act'(z)) (+ res) on the fp32 matrix cores; also gx = gy W
  * sda_row_ln / _bwd     : zuko LayerNorm over the last axis, one 64-lane wavefront per row (shuffle reductions)

Per residual block ``x + Lin2(act(Lin1(LN(x))))`` the forward is three launches (row_ln, linear, linear with the
activation applied to its input and the residual in its epilogue) and saves ``x``, the LN statistics and the
pre-activation ``z``; the input gradient is three launches again (gy W2 * act'(z), gz W1, LN backward + g).
"""
from typing import List

import torch
import torch.nn as nn
from torch import Tensor

from . import ops


def _rows(x: Tensor):
    xs = x.contiguous()
    return xs.reshape(-1, xs.shape[-1])


def _is_res_block(block) -> bool:
    from .nn import LayerNorm
    return (isinstance(block, nn.Sequential) and len(block) == 4 and isinstance(block[0], LayerNorm)
            and isinstance(block[1], nn.Linear) and isinstance(block[3], nn.Linear) and block[0].dim == -1)


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
    out = _MLPFunction.apply(xr, list(layers))
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
