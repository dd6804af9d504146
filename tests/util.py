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
    diff = (a - b).abs()
    err = diff.max().item() if diff.numel() else 0.0
    assert err <= tol, f'{what}: max abs err {err:.3e} > {tol:.3e} (scale {scale:.3e})'
    # and elementwise: |a - b| <= rtol |b| + atol_e with atol_e = rtol/10 x the tensor scale (not below fp32 round-off of
    # that scale), i.e. torch.allclose(a, b, rtol, atol_e)
    atol_e = max(0.1 * rtol, 2e-6) * scale + (atol or 0.0)
    bound = rtol * b.abs() + atol_e
    bad = diff > bound
    if bad.any():
        worst = (diff - bound).argmax().item()
        raise AssertionError(f'{what}: {int(bad.sum())} of {bad.numel()} elements outside rtol={rtol:g}, '
                             f'atol={atol_e:.3e}; worst |a-b|={diff.reshape(-1)[worst]:.3e} '
                             f'at |b|={b.abs().reshape(-1)[worst]:.3e}')


# ---------------------------------------------------------------------------- model builders shared by GPU tests
def build_unet1d_tiny():
    import torch.nn as nn
    from sda_amd.score import MCScoreWrapper, ScoreUNet
    return MCScoreWrapper(ScoreUNet(3, embedding=8, hidden_channels=(8,), hidden_blocks=(1,), activation=nn.SiLU, spatial=1))


def build_unet1d_two_level():
    import torch.nn as nn
    from sda_amd.score import ScoreUNet
    return ScoreUNet(3, embedding=8, hidden_channels=(8, 16), hidden_blocks=(1, 2), activation=nn.SiLU, spatial=1)


def build_mcscore2d_tiny():
    import torch.nn as nn
    from sda_amd.experiments.kolmogorov import LocalScoreUNet
    from sda_amd.score import MCScoreNet
    net = MCScoreNet(2, order=1)
    net.kernel = LocalScoreUNet(channels=6, size=8, embedding=8, hidden_channels=(4, 8), hidden_blocks=(1, 1),
                                kernel_size=3, activation=nn.SiLU, spatial=2, padding_mode='circular')
    return net


def oracle_eps_from_module(module, kind, hidden=(128,) * 5):
    """An oracle (CPU) eps(x, t) sharing the weights of a sda_amd module."""
    from oracle import sda_oracle as O
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    if kind == 'mc2d':
        k = module.kernel
        net = k.network
        cfg = O.UNetConfig(net.in_channels, net.out_channels, net.mod_features, net.hidden_channels, net.hidden_blocks,
                           net.kernel_size[0], net.stride[0], 'SiLU', 2, 'circular')
        order = module.order

        def eps(x, t, dtype=None):
            s = sd if dtype is None else O.cast_sd(sd, dtype)
            kern = lambda xx, tt, c=None: O.score_unet(s, 'kernel.', cfg, xx, tt, s['kernel.forcing'])
            return O.mc_score_net(kern, order, x, t)
        return eps
    if kind == 'wrap1d':
        net = module.score.network
        cfg = O.UNetConfig(net.in_channels, net.out_channels, net.mod_features, net.hidden_channels, net.hidden_blocks,
                           net.kernel_size[0], net.stride[0], 'SiLU', 1, 'zeros')

        def eps(x, t, dtype=None):
            s = sd if dtype is None else O.cast_sd(sd, dtype)
            return O.mc_score_wrapper(lambda xx, tt, c=None: O.score_unet(s, 'score.', cfg, xx, tt, c), x, t)
        return eps
    if kind == 'local':                       # MCScoreNet over a ScoreNet / ResMLP kernel (experiments/lorenz/utils.py:45-59)
        k = module.kernel
        first = k.network[0]                  # Linear(features * window + embedding -> width)
        cfg = O.ResMLPConfig(first.in_features, module.kernel.network[-1][1].in_features if not isinstance(
            k.network[-1], torch.nn.Linear) else k.network[-1].out_features, tuple(hidden), 'SiLU')
        order = module.order

        def eps(x, t, dtype=None):
            s = sd if dtype is None else O.cast_sd(sd, dtype)
            kern = lambda xx, tt, c=None: O.score_net(s, 'kernel.', cfg, xx, tt, c)
            return O.mc_score_net(kern, order, x, t)
        return eps
    raise ValueError(kind)
