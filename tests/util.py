"""Shared helpers for the test-suite (fixtures are data only: tests/golden/*.npz)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    """Returns (arrays, groups): flat entries as tensors, 'group/key' entries nested by group."""
    data = np.load(os.path.join(GOLDEN, name + '.npz'))
    flat, groups = {}, {}
    for k in data.files:
        v = torch.from_numpy(np.asarray(data[k]))
        if '/' in k:
            g, kk = k.split('/', 1)
            groups.setdefault(g, {})[kk] = v
        else:
            flat[k] = v
    return flat, groups


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def assert_close(a, b, rtol=1e-4, atol=None, what=''):
    """max|a-b| <= rtol * max|b| (+atol): the scale-relative form of north_star's rtol=1e-4 (fp32)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, f'{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    scale = b.abs().max().item()
    tol = rtol * scale + (atol or 0.0)
    err = (a - b).abs().max().item()
    assert err <= tol, f'{what}: max abs err {err:.3e} > {tol:.3e} (scale {scale:.3e})'
