r"""Kolmogorov score-network factories (experiments/kolmogorov/utils.py:29-81 of the reference)."""

from pathlib import Path
from typing import Sequence

import torch
import torch.nn as nn
from torch import Tensor

from ..score import MCScoreNet, ScoreUNet
from ..utils import ACTIVATIONS, load_config


class LocalScoreUNet(ScoreUNet):
    r"""Score U-Net with a forcing channel sin(4 * 2 pi (i + 1/2) / size) as its single context channel
    (kolmogorov/utils.py:29-46), defined as the reference defines it: ``forward`` hands ``self.forcing`` on as the context.
    The channel is never concatenated (the head convolution's loader reads it as a broadcast context plane), and inside an
    ``MCScoreNet`` the override is recognised as context-only (``score._context_only_override``), so this class -- and the
    reference's own, when its driver file runs unchanged on this package -- takes the fused window path."""

    def __init__(self, channels: int, size: int = 64, **kwargs):
        super().__init__(channels, 1, **kwargs)
        domain = 2 * torch.pi / size * (torch.arange(size) + 1 / 2)
        self.register_buffer('forcing', torch.sin(4 * domain).expand(1, size, size).clone())

    def forward(self, x: Tensor, t: Tensor, c: Tensor = None) -> Tensor:
        return super().forward(x, t, self.forcing)


def make_score(
    window: int = 3,
    embedding: int = 64,
    hidden_channels: Sequence[int] = (64, 128, 256),
    hidden_blocks: Sequence[int] = (3, 3, 3),
    kernel_size: int = 3,
    activation: str = 'SiLU',
    size: int = 64,
    **absorb,
) -> nn.Module:
    score = MCScoreNet(2, order=window // 2)
    score.kernel = LocalScoreUNet(
        channels=window * 2,
        size=size,
        embedding=embedding,
        hidden_channels=hidden_channels,
        hidden_blocks=hidden_blocks,
        kernel_size=kernel_size,
        activation=ACTIVATIONS[activation],
        spatial=2,
        padding_mode='circular',
    )
    return score


def load_score(file: Path, device: str = 'cpu', **kwargs) -> nn.Module:
    state = torch.load(file, map_location=device)
    config = load_config(Path(file).parent)
    config.update(kwargs)
    score = make_score(**config)
    score.load_state_dict(state)
    return score
