#!/usr/bin/env python3 -B
"""Golden fixtures for the evaluation metrics, generated FROM THE REFERENCE's sda/utils.py.

Runs only in the build container (needs /root/reference).  `sda/utils.py` imports h5py, POT (`ot`) and tqdm at module
level; h5py/POT are absent here and neither is touched by `mmd` or `bpf`, so empty module objects are registered under
those names purely so that the file can be executed.  `emd` calls `ot.emd2` and therefore CANNOT be run here: its fixture
is absent and the emd oracle is "parity unpinned" (oracle/sda_oracle.py:emd).

Usage:  python3 -B tests/golden/make_golden_metrics.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as G  # noqa: E402  (zuko stand-in + reference loader)


def main():
    G._load_reference()
    for name in ('h5py', 'ot'):
        try:
            __import__(name)
        except ImportError:
            sys.modules[name] = types.ModuleType(name)
    spec = importlib.util.spec_from_file_location('sda.utils', os.path.join(G.REF, 'sda', 'utils.py'))
    rutils = importlib.util.module_from_spec(spec)
    sys.modules['sda.utils'] = rutils
    spec.loader.exec_module(rutils)

    # ---------------------------------------------------------------- mmd: two sample sets of different size
    torch.manual_seed(21)
    x = torch.randn(48, 5, 3) * 0.7
    y = torch.randn(40, 5, 3) * 0.9 + 0.2
    G._save('metrics_mmd', x=x, y=y, mmd_xy=rutils.mmd(x, y), mmd_xx=rutils.mmd(x, x[:24]))

    # ---------------------------------------------------------------- bpf: noisy rotation, Gaussian likelihood on x[0]
    torch.manual_seed(22)
    x0 = torch.randn(64, 2)
    yobs = torch.randn(5, 1)
    rot = torch.tensor([[0.96, -0.28], [0.28, 0.96]])

    def transition(x):
        return x @ rot.T + 0.1 * torch.randn_like(x)

    def likelihood(yi, x):
        return torch.softmax(-((x[:, :1] - yi) ** 2).sum(-1) / 0.5, 0)

    torch.manual_seed(23)
    out = rutils.bpf(x0, yobs, transition, likelihood, step=2)
    G._save('metrics_bpf', x0=x0, y=yobs, rot=rot, seed=np.array(23), step=np.array(2), out=out)


if __name__ == '__main__':
    main()
