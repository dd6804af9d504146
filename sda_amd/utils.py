r"""Helpers on the sampling path (subset of the reference's ``sda/utils.py``: ACTIVATIONS and the run-config reader,
sda/utils.py:19-25,40-42; the evaluation metrics ``bpf`` / ``emd`` / ``mmd`` of sda/utils.py:168-263 live in
``sda_amd.metrics`` and are re-exported here under the reference's names).  Training loop, datasets and the config
writers are out of scope.

``from sda.utils import *`` in the reference's drivers also hands on that module's own imports (sda/utils.py:3-16:
``json``, ``math``, ``torch``, ``Path``, ``Tensor``, the ``typing`` names, everything of ``sda.score`` and, where installed,
``h5py`` -- experiments/lorenz/eval.py reads its observations with it); the same names are importable from here."""

import json
import math  # noqa: F401
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
