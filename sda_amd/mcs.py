r"""``sda.mcs`` for drivers that run on this package -- a passthrough, not an implementation.

The reference's experiment helpers start with ``from sda.mcs import *`` (experiments/lorenz/utils.py:8,
experiments/kolmogorov/utils.py:11).  The simulators of ``sda/mcs.py`` are data generation, outside the sampling hot path
(SURVEY.md section 2), so nothing of them is rebuilt here.  This module

* re-exports the user's own ``sda/mcs.py`` when one is installed next to this package (first ``sda`` package on ``sys.path``
  that is not this one, or the file named by ``$SDA_MCS_FILE``) and its imports (jax, optionally jax_cfd) succeed;
* otherwise exports the names a driver needs at import time -- the module-level imports of sda/mcs.py:7-19 that the drivers
  pick up through the star import (``np``, ``math``, ``torch``, ``Tensor``, ``Size``, ``Normal``, ``MultivariateNormal``,
  ``typing``), ``MarkovChain`` (an annotation in ``make_chain``) and the chain classes as placeholders that raise a clear
  ``ImportError`` when instantiated.  The three torch-only static helpers observation operators are written with
  (``KolmogorovFlow.coarsen / upsample / vorticity``, sda/mcs.py:340-375) work in both modes: they are plain torch, so any
  ``A`` built from them differentiates through autograd exactly as with the reference (``sda_amd.observe`` has the same
  operators with hand-written adjoints).

No relative imports on purpose: the module is also importable as ``sda.mcs`` through ``sys.modules['sda'] = sda_amd``.
"""

import abc
import importlib.machinery
import importlib.util
import math  # noqa: F401  (re-exported, as sda/mcs.py:8)
import os
import random  # noqa: F401
import sys

import numpy as np  # noqa: F401
import torch
from torch import Size, Tensor  # noqa: F401
from torch.distributions import MultivariateNormal, Normal  # noqa: F401
from typing import *  # noqa: F401,F403

#: where the chains came from: the path of the user's sda/mcs.py, or None (placeholders)
SOURCE = None
#: why the passthrough did not happen (None when it did)
UNAVAILABLE = None


def _find_user_mcs():
    path = os.environ.get('SDA_MCS_FILE')
    if path:
        return path
    here = os.path.dirname(os.path.abspath(__file__))
    for entry in list(sys.path):
        spec = importlib.machinery.PathFinder.find_spec('sda', [entry or os.getcwd()])
        if spec is None or not spec.submodule_search_locations:
            continue
        for loc in spec.submodule_search_locations:
            cand = os.path.join(loc, 'mcs.py')
            if os.path.abspath(loc) != here and os.path.exists(cand):
                return cand
    return None


def _passthrough():
    global SOURCE, UNAVAILABLE
    path = _find_user_mcs()
    if path is None:
        UNAVAILABLE = ('no sda/mcs.py found on sys.path (install the reference package next to sda_amd, or point '
                       '$SDA_MCS_FILE at its mcs.py)')
        return {}
    spec = importlib.util.spec_from_file_location('sda_amd._user_mcs', path)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except Exception as e:  # noqa: BLE001 -- jax (or another simulator dependency) is missing (ImportError) or mismatched: a jax / jaxlib
        # version skew raises RuntimeError / AttributeError at import time.  Drivers that only SAMPLE never touch a simulator, so
        # install_as_sda() must survive any of them; the placeholders below say what happened when a chain is instantiated
        UNAVAILABLE = f'{path} could not be imported: {type(e).__name__}: {e}'
        return {}
    SOURCE = path
    return {k: v for k, v in vars(mod).items() if not k.startswith('_')}


globals().update(_passthrough())

if SOURCE is None:

    class MarkovChain(abc.ABC):
        """Placeholder for sda/mcs.py:22-57 (abstract first-order Markov chain); see the module docstring."""

        def __init__(self, *args, **kwargs):
            raise ImportError(f'{type(self).__name__} is a simulator of the reference package (sda/mcs.py), which sda_amd does not '
                              f'rebuild -- it accelerates posterior sampling only.  {UNAVAILABLE}')

    def _placeholder(name: str, **statics):
        return type(name, (MarkovChain,), {'__doc__': f'Placeholder for sda.mcs.{name}.', '__module__': __name__, **statics})

    def _coarsen(x: Tensor, r: int = 2) -> Tensor:
        """Mean over r x r cells of the last two axes (sda/mcs.py:340-347)."""
        h, w = x.shape[-2:]
        return x.unflatten(-1, (w // r, r)).unflatten(-3, (h // r, r)).mean(dim=(-3, -1))

    def _upsample(x: Tensor, r: int = 2, mode: str = 'bilinear') -> Tensor:
        """Periodic interpolation by a factor r (sda/mcs.py:349-359): one wrapped cell on each side, interpolate, crop."""
        h, w = x.shape[-2:]
        y = torch.nn.functional.pad(x.reshape(-1, 1, h, w), (1, 1, 1, 1), mode='circular')
        y = torch.nn.functional.interpolate(y, scale_factor=(r, r), mode=mode)
        return y[..., r:-r, r:-r].reshape(*x.shape[:-2], r * h, r * w)

    def _vorticity(x: Tensor) -> Tensor:
        """d u / d x - d v / d y by periodic central differences of a (..., 2, H, W) velocity field (sda/mcs.py:361-375)."""
        u, v = x[..., 0, :, :], x[..., 1, :, :]
        du = (torch.roll(u, -1, -1) - torch.roll(u, 1, -1)) / 2
        dv = (torch.roll(v, -1, -2) - torch.roll(v, 1, -2)) / 2
        return du - dv

    DampedSpring = _placeholder('DampedSpring')
    DiscreteODE = _placeholder('DiscreteODE')
    Lorenz63 = _placeholder('Lorenz63')
    NoisyLorenz63 = _placeholder('NoisyLorenz63')
    Lorenz96 = _placeholder('Lorenz96')
    LotkaVolterra = _placeholder('LotkaVolterra')
    KolmogorovFlow = _placeholder('KolmogorovFlow', coarsen=staticmethod(_coarsen), upsample=staticmethod(_upsample),
                                  vorticity=staticmethod(_vorticity))
