"""ResMLP / ScoreNet path (the Lorenz *local* kernel; sda/nn.py:31-71, sda/score.py:38-63) on MI355X.

Row-major ``(rows, features)`` activations; kernels in csrc/linear.hip:
  * sda_linear      : Y = act(X W^T + b) (+ residual) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32)
  * sda_row_ln      : zuko LayerNorm over the last axis, one 64-lane wavefront per row (shuffle reductions)
"""
import torch
from torch import Tensor

from . import ops
from ._lib import SdaHipError


def _rows(x: Tensor):
    xs = x.contiguous()
    return xs.reshape(-1, xs.shape[-1])


def row_layer_norm(x: Tensor, eps: float, unbiased: bool) -> Tensor:
    ops._dev(x)
    xr = _rows(x)
    y = torch.empty_like(xr)
    ops.row_ln(xr, eps, unbiased, y)
    return y.reshape(x.shape)


def try_fused_residual_mlp(block, x: Tensor):
    """ResidualBlock(LayerNorm(), Linear, act, Linear) -> x + Lin(act(Lin(LN(x)))) with bias/act/residual fused."""
    from .nn import LN_UNBIASED, LayerNorm, activation_id
    import torch.nn as nn
    if not (len(block) == 4 and isinstance(block[0], LayerNorm) and isinstance(block[1], nn.Linear)
            and isinstance(block[3], nn.Linear) and block[0].dim in (-1,)):
        return None
    ops._dev(x)
    act = activation_id(block[2])
    xr = _rows(x)
    h = torch.empty_like(xr)
    ops.row_ln(xr, block[0].eps, LN_UNBIASED, h)
    h1 = ops.linear(h, block[1].weight.detach(), block[1].bias, act=act)
    out = ops.linear(h1, block[3].weight.detach(), block[3].bias, act=0, res=xr)
    return out.reshape(x.shape)


def resmlp_forward(mlp, x: Tensor) -> Tensor:
    import torch.nn as nn
    ops._dev(x)
    shape = x.shape
    h = _rows(x)
    for layer in mlp:
        if isinstance(layer, nn.Linear):
            h = ops.linear(h, layer.weight.detach(), layer.bias, act=0)
        else:
            h = layer(h)
    return h.reshape(*shape[:-1], h.shape[-1])
