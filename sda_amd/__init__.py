r"""sda_amd -- MI355X-native posterior-sampling hot path of Score-based Data Assimilation (francois-rozet/sda).

``import sda_amd as sda`` exposes the reference's ``sda.nn`` / ``sda.score`` / ``sda.utils`` surface for that
path; the arithmetic runs in hand-written gfx950 kernels (``sda_amd/csrc`` -> ``lib/libsda_hip.so``, C ABI in
``include/sda_hip.h``).  There is no CPU fallback.
"""

from . import nn
from . import score
from . import utils
from . import observe

__version__ = '0.1.0'
