r"""Helpers on the sampling path (subset of the reference's ``sda/utils.py``: ACTIVATIONS and the run-config reader,
sda/utils.py:19-25,40-42; the evaluation metrics ``bpf`` / ``emd`` / ``mmd`` of sda/utils.py:168-263 live in
``sda_amd.metrics`` and are re-exported here under the reference's names; the two config writers the reference's
``experiments/*/train.py`` reach through ``from sda.utils import *`` -- ``random_config`` / ``save_config``, sda/utils.py:28-37 --
are plain host code and kept).  Training loop and datasets are out of scope.

``from sda.utils import *`` in the reference's drivers also hands on that module's own imports (sda/utils.py:3-16:
``json``, ``math``, ``torch``, ``Path``, ``Tensor``, the ``typing`` names, everything of ``sda.score`` and, where installed,
``h5py`` -- experiments/lorenz/eval.py reads its observations with it); the same names are importable from here."""

import json
import math  # noqa: F401
import random
from pathlib import Path
from typing import *  # noqa: F401,F403

import torch
from torch import Tensor  # noqa: F401

try:
    import h5py  # noqa: F401  (optional: only the drivers' dataset / observation files need it)
except ImportError:
    pass

from .score import *  # noqa: F401,F403  (sda/utils.py:16)
from .metrics import bpf, emd, mmd  # noqa: F401  (sda.utils.bpf / emd / mmd)

ACTIVATIONS = {
    'ReLU': torch.nn.ReLU,
    'ELU': torch.nn.ELU,
    'GELU': torch.nn.GELU,
    'SELU': torch.nn.SELU,
    'SiLU': torch.nn.SiLU,
}


def load_config(path: Path) -> Dict[str, Any]:
    with open(Path(path) / 'config.json', mode='r') as f:
        return json.load(f)


def save_config(config: Dict[str, Any], path: Path) -> None:
    """Write ``path/config.json``; refuses to overwrite an existing run's file (exclusive create, as sda/utils.py:35-37)."""
    with open(Path(path) / 'config.json', mode='x') as f:
        json.dump(config, f)


def random_config(configs: Dict[str, Sequence[Any]]) -> Dict[str, Any]:
    """One uniformly drawn value per key of a {key: candidates} search space (sda/utils.py:28-32; python's ``random``)."""
    drawn = {}
    for key, values in configs.items():
        drawn[key] = random.choice(values)
    return drawn
