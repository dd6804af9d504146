"""TEST-ONLY (build container): load the reference's experiment driver files UNMODIFIED on top of sda_amd.

`/root/reference/experiments/{lorenz,kolmogorov}/utils.py` start with ``from sda.mcs import *; from sda.score import *;
from sda.utils import *`` -- `driver(name)` makes those resolve to this package (``sda_amd.install_as_sda()``), stubs the
two plotting dependencies the Kolmogorov helper imports when they are absent (seaborn; PIL is present here), executes the
file from where it lies (never copied, no bytecode written) in a scratch working directory (the helpers ``mkdir`` their
``PATH``), and restores ``sys.modules`` / the working directory afterwards.  Nothing here runs on the GPU box:
`/root/reference` does not exist there, and the tests that use this skip without it.
"""
import contextlib
import importlib.util
import os
import sys
import tempfile
import types

REF = '/root/reference'
DRIVERS = {'lorenz': 'experiments/lorenz/utils.py', 'kolmogorov': 'experiments/kolmogorov/utils.py'}


def have_reference() -> bool:
    return all(os.path.exists(os.path.join(REF, p)) for p in DRIVERS.values())


def _stub_if_missing(name: str, **attrs):
    try:
        importlib.import_module(name)
        return None
    except ImportError:
        mod = types.ModuleType(name)
        mod.__dict__.update(attrs)
        sys.modules[name] = mod
        return name


def exec_driver(name: str, modname: str):
    """Execute one of the reference's driver files as module `modname` (whatever `sda` currently resolves to)."""
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, DRIVERS[name]))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@contextlib.contextmanager
def scratch_cwd():
    keep = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            yield tmp
        finally:
            os.chdir(keep)


@contextlib.contextmanager
def driver(name: str):
    """The reference's `experiments/<name>/utils.py` running on sda_amd; yields the module."""
    import sda_amd
    keep_mods = {k: v for k, v in sys.modules.items() if k == 'sda' or k.startswith('sda.')}
    keep_flag = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    stubs = []
    try:
        sda_amd.install_as_sda()
        stubs = [s for s in (_stub_if_missing('seaborn', cm=types.SimpleNamespace(icefire=lambda w: w)),) if s]
        with scratch_cwd():
            yield exec_driver(name, f'_reference_{name}_utils')
    finally:
        sys.dont_write_bytecode = keep_flag
        for k in [k for k in sys.modules if k == 'sda' or k.startswith('sda.')]:
            del sys.modules[k]
        sys.modules.update(keep_mods)
        for s in stubs:
            sys.modules.pop(s, None)
        sys.modules.pop(f'_reference_{name}_utils', None)
