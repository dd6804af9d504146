r"""sda_amd -- MI355X-native posterior-sampling hot path of Score-based Data Assimilation (francois-rozet/sda).

``import sda_amd as sda`` exposes the reference's ``sda.nn`` / ``sda.score`` / ``sda.utils`` surface for that
path; the arithmetic runs in hand-written gfx950 kernels (``sda_amd/csrc`` -> ``lib/libsda_hip.so``, C ABI in
``include/sda_hip.h``).  There is no CPU fallback.
"""

import importlib
import sys

from . import nn
from . import score
from . import utils
from . import observe

__version__ = '0.1.0'


def __getattr__(name):
    # ``sda_amd.mcs`` loads on first use: it probes for the user's own sda/mcs.py (and with it jax), see mcs.py
    if name == 'mcs':
        return importlib.import_module('.mcs', __name__)
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')


def install_as_sda() -> None:
    """Make ``import sda`` / ``from sda.{mcs,nn,score,utils} import *`` resolve to this package, so that the reference's driver
    files (experiments/lorenz/utils.py:8-10, experiments/kolmogorov/utils.py:11-13, lorenz/eval.py:9-11) run unchanged:

        import sda_amd; sda_amd.install_as_sda()
        from utils import *            # the reference's experiments/<name>/utils.py, unmodified

    Every submodule is registered under its ``sda.`` name explicitly -- a bare ``sys.modules['sda'] = sda_amd`` would make
    the import system load second copies of the submodules (distinct classes) on ``import sda.<name>``."""
    me = sys.modules[__name__]
    sys.modules['sda'] = me
    # every public submodule is imported BEFORE aliasing: a later `import sda.parallel` must find the one copy, not load a second
    for sub in ('mcs', 'nn', 'score', 'utils', 'observe', 'parallel', 'metrics', 'ops', 'engine', 'mlp', 'fused1d', 'experiments',
                'experiments.kolmogorov', 'experiments.lorenz'):
        importlib.import_module('.' + sub, __name__)
    for full, mod in list(sys.modules.items()):
        if full.startswith(__name__ + '.') and mod is not None:
            sys.modules['sda.' + full[len(__name__) + 1:]] = mod
