r"""Neural networks -- MI355X host-side mirror of the reference's ``sda/nn.py``.

Same public classes, constructor signatures and ``state_dict`` key names as the reference
(``UNet``, ``ModResidualBlock``, ``ResMLP``, ``ResidualBlock``; sda/nn.py:11-206), so reference
checkpoints load unchanged.  The ``nn.Module`` tree here only *owns the
parameters*; the arithmetic runs in hand-written gfx950 kernels through :mod:`sda_amd.engine`
(implicit-GEMM convolutions on the fp32 matrix cores with LayerNorm / modulation / activation /
upsample fused into their loaders).  There is no CPU path: CPU tensors raise ``SdaHipError``.

``LayerNorm`` stands in for ``zuko.nn.LayerNorm`` (zuko==0.1.4, environment.yml:23), restated from
its published definition: unbiased variance, ``eps=1e-5``, no affine parameters.
"""

from typing import Callable, Iterable, Sequence, Union

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from ._lib import ACT_IDS, SdaHipError

LN_UNBIASED = True       # zuko 0.1.4 convention; the single switch (mirrors oracle/sda_oracle.py)

_ACT_NAMES = {nn.ReLU: 'ReLU', nn.ELU: 'ELU', nn.GELU: 'GELU', nn.SELU: 'SELU', nn.SiLU: 'SiLU', nn.Identity: None}


def activation_id(act: nn.Module) -> int:
    for cls, name in _ACT_NAMES.items():
        if type(act) is cls:
            if isinstance(act, nn.ELU) and act.alpha != 1.0:
                break
            if isinstance(act, nn.GELU) and act.approximate != 'none':
                break
            return ACT_IDS[name]
    raise NotImplementedError(f'activation {act!r} has no gfx950 kernel (supported: ReLU, ELU, GELU, SELU, SiLU)')


class LayerNorm(nn.Module):
    r"""Standardises features along one dimension, :math:`(x - E[x]) / \sqrt{V[x] + \epsilon}`; no affine.

    On MI355X two layouts are implemented: the channel axis of a planar ``(N, C, *spatial)`` tensor
    (``dim = -(spatial+1)``, the U-Net case, nn.py:137,163) and the last axis (``dim=-1``, ResMLP, nn.py:61).
    """

    def __init__(self, dim: Union[int, Iterable[int]] = -1, eps: float = 1e-5):
        super().__init__()
        self.dim = dim if type(dim) is int else tuple(dim)
        self.eps = eps

    def extra_repr(self) -> str:
        return f'dim={self.dim}'

    def forward(self, x: Tensor) -> Tensor:
        if type(self.dim) is not int:
            raise NotImplementedError('LayerNorm over several dims has no gfx950 kernel')
        d = self.dim % x.dim()
        if d == x.dim() - 1:
            from . import mlp
            return mlp.row_layer_norm(x, self.eps, LN_UNBIASED)
        if d < 1:
            raise NotImplementedError('channel LayerNorm needs a leading batch axis')
        # planar channel norm: collapse batch dims before d, spatial dims after
        xs = x.contiguous()
        n = int(torch.tensor(xs.shape[:d]).prod()) if d > 0 else 1
        c = xs.shape[d]
        xv = xs.reshape(n, c, -1)
        mean = torch.empty(n * xv.shape[2], device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        ops.ln_stats(xv, None, 0, self.eps, LN_UNBIASED, mean, rstd)
        y = torch.empty_like(xv)
        ops.ln_apply(xv, None, 0, mean, rstd, y)
        return y.reshape(x.shape)


class ResidualBlock(nn.Sequential):
    r"""x + f(x) for a sequential f."""

    def forward(self, x: Tensor) -> Tensor:
        from . import mlp
        fused = mlp.try_fused_residual_mlp(self, x)
        if fused is not None:
            return fused
        return x + super().forward(x)


class ModResidualBlock(nn.Module):
    r"""x + residue(x + project(y)): residual block with additive modulation (nn.py:18-28)."""

    def __init__(self, project: nn.Module, residue: nn.Module):
        super().__init__()
        self.project = project
        self.residue = residue

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        from .engine import block_forward_standalone
        return block_forward_standalone(self, x, y)


class ResMLP(nn.Sequential):
    r"""Residual MLP: per width step an optional ``Linear`` then ``x + Lin(act(Lin(LN(x))))`` (nn.py:31-71)."""

    def __init__(
        self,
        in_features: int,
        out_features: int,
        hidden_features: Sequence[int] = (64, 64),
        activation: Callable[[], nn.Module] = nn.ReLU,
        **kwargs,
    ):
        widths = [in_features, *hidden_features, out_features]
        layers = []
        for prev, cur in zip(widths[:-1], widths[1:]):
            if prev != cur:
                layers.append(nn.Linear(prev, cur, **kwargs))
            layers.append(ResidualBlock(LayerNorm(), nn.Linear(cur, cur, **kwargs), activation(),
                                        nn.Linear(cur, cur, **kwargs)))
        super().__init__(*layers)
        self.in_features = in_features
        self.out_features = out_features

    def forward(self, x: Tensor) -> Tensor:
        from . import mlp
        return mlp.resmlp_forward(self, x)


_CONVS = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}


class UNet(nn.Module):
    r"""U-Net with additive time modulation (nn.py:74-206).

    Arguments are the reference's: ``in_channels, out_channels, mod_features, hidden_channels,
    hidden_blocks, kernel_size, stride, activation, spatial`` and ``**kwargs`` forwarded to the
    convolutions (``padding_mode='circular'`` for Kolmogorov).  ``spatial`` 1 and 2 run on the tuned kernels, ``spatial=3``
    on the general 3-D kernel (``engine3d``).

    Parameter layout (matches the reference's state_dict): ``heads.0`` / ``tails.{D-1}`` are plain
    convolutions, deeper heads are ``Sequential(conv)``, deeper tails ``Sequential(LayerNorm,
    Upsample, conv)``; ``tails`` and ``ascent`` are stored deepest level first.
    """

    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        mod_features: int,
        hidden_channels: Sequence[int] = (32, 64, 128),
        hidden_blocks: Sequence[int] = (2, 3, 5),
        kernel_size: Union[int, Sequence[int]] = 3,
        stride: Union[int, Sequence[int]] = 2,
        activation: Callable[[], nn.Module] = nn.ReLU,
        spatial: int = 2,
        **kwargs,
    ):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.mod_features = mod_features
        self.spatial = spatial
        self.hidden_channels = tuple(hidden_channels)
        self.hidden_blocks = tuple(hidden_blocks)

        conv = _CONVS[spatial]
        ksize = [kernel_size] * spatial if type(kernel_size) is int else list(kernel_size)
        strides = [stride] * spatial if type(stride) is int else list(stride)
        self.kernel_size = tuple(ksize)
        self.stride = tuple(strides)
        conv_kw = dict(kwargs, kernel_size=ksize, padding=[k // 2 for k in ksize])

        def make_block(width: int) -> ModResidualBlock:
            project = nn.Sequential(nn.Linear(mod_features, width), nn.Unflatten(-1, (-1,) + (1,) * spatial))
            residue = nn.Sequential(LayerNorm(-(spatial + 1)), conv(width, width, **conv_kw), activation(),
                                    conv(width, width, **conv_kw))
            return ModResidualBlock(project=project, residue=residue)

        depth = len(self.hidden_blocks)
        heads, tails, descent, ascent = [], [], [], []
        for lvl in range(depth):
            width = self.hidden_channels[lvl]
            if lvl == 0:
                heads.append(conv(in_channels, width, **conv_kw))
                tails.append(conv(width, out_channels, **conv_kw))
            else:
                below = self.hidden_channels[lvl - 1]
                heads.append(nn.Sequential(conv(below, width, stride=strides, **conv_kw)))
                tails.append(nn.Sequential(LayerNorm(-(spatial + 1)),
                                           nn.Upsample(scale_factor=tuple(strides), mode='nearest'),
                                           conv(width, below, **conv_kw)))
            descent.append(nn.ModuleList(make_block(width) for _ in range(self.hidden_blocks[lvl])))
            ascent.append(nn.ModuleList(make_block(width) for _ in range(self.hidden_blocks[lvl])))

        self.heads = nn.ModuleList(heads)
        self.tails = nn.ModuleList(tails[::-1])
        self.descent = nn.ModuleList(descent)
        self.ascent = nn.ModuleList(ascent[::-1])
        self._engine = None

    # -- engine plumbing ---------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            if self.spatial == 3:
                from .engine3d import UNet3dEngine as UNetEngine
            else:
                from .engine import UNetEngine
            object.__setattr__(self, '_engine', UNetEngine(self))
        return self._engine

    def invalidate_engine(self):
        r"""Drop the engine's packed-weight caches.  They follow parameter updates on their own (pointer + version keys:
        optimiser steps, ``.to()``, ``load_state_dict``); only writes through ``.data`` -- e.g. an EMA swap
        ``p.data.copy_(ema)`` -- need this call."""
        if self._engine is not None:
            self._engine.invalidate()

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        r"""x: ``(N, in_channels, *spatial)``; y: ``(N | 1, mod_features)`` -> ``(N, out_channels, *spatial)``."""
        from .engine import unet_apply
        return unet_apply(self, x, y)
