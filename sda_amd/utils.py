r"""Helpers on the sampling path (subset of the reference's ``sda/utils.py``: ACTIVATIONS and the run config
json, sda/utils.py:19-42; the evaluation metrics ``bpf`` / ``emd`` / ``mmd`` of sda/utils.py:168-263 live in
``sda_amd.metrics`` and are re-exported here under the reference's names).  Training loop and datasets are out of scope
(SURVEY.md section 2)."""

import json
import random
from pathlib import Path
from typing import Any, Dict, Sequence

import torch

from .metrics import bpf, emd, mmd  # noqa: F401  (sda.utils.bpf / emd / mmd)

ACTIVATIONS = {
    'ReLU': torch.nn.ReLU,
    'ELU': torch.nn.ELU,
    'GELU': torch.nn.GELU,
    'SELU': torch.nn.SELU,
    'SiLU': torch.nn.SiLU,
}


def random_config(configs: Dict[str, Sequence[Any]]) -> Dict[str, Any]:
    return {key: random.choice(values) for key, values in configs.items()}


def save_config(config: Dict[str, Any], path: Path) -> None:
    with open(Path(path) / 'config.json', mode='x') as f:
        json.dump(config, f)


def load_config(path: Path) -> Dict[str, Any]:
    with open(Path(path) / 'config.json', mode='r') as f:
        return json.load(f)
