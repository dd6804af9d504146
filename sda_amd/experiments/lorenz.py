r"""Lorenz score-network factories (experiments/lorenz/utils.py:26-79 of the reference)."""

from pathlib import Path
from typing import Sequence

import torch
import torch.nn as nn

from ..score import MCScoreNet, MCScoreWrapper, ScoreUNet
from ..utils import ACTIVATIONS, load_config


def make_global_score(
    embedding: int = 32,
    hidden_channels: Sequence[int] = (64,),
    hidden_blocks: Sequence[int] = (3,),
    activation: str = 'SiLU',
    channels: int = 3,
    **absorb,
) -> nn.Module:
    return MCScoreWrapper(ScoreUNet(channels=channels, embedding=embedding, hidden_channels=hidden_channels,
                                    hidden_blocks=hidden_blocks, activation=ACTIVATIONS[activation], spatial=1))


def make_local_score(
    window: int = 5,
    embedding: int = 32,
    width: int = 128,
    depth: int = 5,
    activation: str = 'SiLU',
    features: int = 3,
    **absorb,
) -> nn.Module:
    return MCScoreNet(features=features, order=window // 2, embedding=embedding, hidden_features=[width] * depth,
                      activation=ACTIVATIONS[activation])


def load_score(file: Path, local: bool = False, device: str = 'cpu', **kwargs) -> nn.Module:
    state = torch.load(file, map_location=device)
    config = load_config(Path(file).parent)
    config.update(kwargs)
    score = make_local_score(**config) if local else make_global_score(**config)
    score.load_state_dict(state)
    return score
