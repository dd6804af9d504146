#!/usr/bin/env python3
"""Randomised parity sweep of the guided-score path: GaussianScore(MCScoreNet) on the GPU -- with the engine forced to chunk
the windows, to keep only part of the activations (recompute path) and to stream the batch in groups -- against the float64
oracle's gaussian_score; and the Lorenz local path (MCScoreNet over a ResMLP kernel), forward + VJP.

    python tests/fuzz/path_fuzz.py [--cases 40] [--seed 0]
"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import sda_oracle as O  # noqa: E402
from sda_amd import engine as E  # noqa: E402
from sda_amd.score import GaussianScore, MCScoreNet, VPSDE  # noqa: E402

SMOOTH = {'SiLU': nn.SiLU, 'GELU': nn.GELU, 'ELU': nn.ELU}     # smooth activations: no kink-flip ill-conditioning


def check(name, got, want, cfg):
    got, want = got.detach().cpu().double(), want.detach().double()
    if not torch.isfinite(got).all():
        return f'{name}: non-finite'
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    return None if err <= 1e-4 * scale else f'{name}: max abs err {err:.3e} vs scale {scale:.3e}'


def guided_case(rng, dev, idx):
    spatial = rng.choice([1, 2])
    depth = rng.choice([1, 2])
    c0 = rng.choice([4, 8, 96]) if spatial == 2 else rng.choice([8, 64])
    hidden = tuple(c0 * 2 ** i for i in range(depth))
    blocks = tuple(rng.choice([1, 2]) for _ in range(depth))
    act = rng.choice(list(SMOOTH))
    pad = rng.choice(['zeros', 'circular'])
    state, order = rng.choice([1, 2, 3]), rng.choice([1, 2])
    size = [2 ** (depth - 1) * rng.choice([2, 4, 8]) for _ in range(spatial)]
    if c0 == 96:
        size = [min(s, 8) for s in size]
    B, L = rng.choice([2, 3, 5]), 2 * order + rng.choice([1, 3, 6])
    emb = 8
    mode = rng.choice(['plain', 'chunk1', 'keep_some', 'groups'])
    y_per_row = rng.random() < 0.5
    cfg = dict(kind='guided', spatial=spatial, hidden=hidden, blocks=blocks, act=act, pad=pad, state=state, order=order,
               size=size, B=B, L=L, mode=mode, y_per_row=y_per_row)
    torch.manual_seed(4000 + idx)
    net = MCScoreNet(state, order=order, embedding=emb, hidden_channels=hidden, hidden_blocks=blocks, kernel_size=3,
                     activation=SMOOTH[act], spatial=spatial, padding_mode=pad)
    ocfg = O.UNetConfig(state * (2 * order + 1), state * (2 * order + 1), emb, hidden, blocks, 3, 2, act, spatial, pad)
    sd = {k: v.detach().double() for k, v in net.state_dict().items()}
    x = torch.randn(B, L, state, *size)
    t = torch.rand(()) * 0.8 + 0.1
    step = rng.choice([1, 2])
    A = (lambda v: v[..., ::step, :1, ::2]) if spatial == 1 else (lambda v: v[..., ::step, :, ::2, ::2])
    y = torch.randn(A(x).shape if y_per_row else A(x[0]).shape)
    std, gamma = 0.3, 3e-2

    def eps64(xx, tt):
        return O.mc_score_net(lambda a, b, c=None: O.score_unet(sd, 'kernel.', ocfg, a, b, None), order, xx, tt)
    want = O.gaussian_score(eps64, O.Schedule('cos'), y.double(), A, std, gamma, x.double(), t.double())

    net = net.to(dev)
    gs = GaussianScore(y, A=A, std=std, sde=VPSDE(net, shape=()), gamma=gamma).to(dev)
    saved = (E.CHUNK_HBM_FRACTION, E.KEEP_HBM_FRACTION, E.UNetEngine.chunk_size)
    try:
        nwin = B * (L - 2 * order)
        if mode == 'chunk1':             # nothing kept, one window per chunk: pure recompute path
            E.UNetEngine.chunk_size = lambda self, n, hs, ws, save, device, fraction=None: 1
        elif mode == 'keep_some':        # keep a leading part (>= 8 windows or n//16), recompute the rest in chunks of 3
            E.UNetEngine.chunk_size = lambda self, n, hs, ws, save, device, fraction=None: \
                (min(n, max(8, nwin // 2)) if fraction is not None else min(n, 3))
        elif mode == 'groups':
            gs.group_size = rng.choice([1, 2])
        got = gs(x.to(dev), t.to(dev))
    except Exception as e:  # noqa: BLE001
        return cfg, f'EXCEPTION {type(e).__name__}: {e}'
    finally:
        E.CHUNK_HBM_FRACTION, E.KEEP_HBM_FRACTION, E.UNetEngine.chunk_size = saved
    return cfg, check('guided', got, want, cfg)


def local_case(rng, dev, idx):
    state, order = rng.choice([1, 3, 5, 40]), rng.choice([1, 2, 4])
    widths = [rng.choice([16, 64, 100, 256])] * rng.choice([1, 2, 3])
    act = rng.choice(list(SMOOTH))
    emb = rng.choice([8, 32])
    B, L = rng.choice([1, 3, 64]), 2 * order + rng.choice([1, 5, 30])
    cfg = dict(kind='local', state=state, order=order, widths=widths, act=act, emb=emb, B=B, L=L)
    torch.manual_seed(6000 + idx)
    net = MCScoreNet(features=state, order=order, embedding=emb, hidden_features=widths, activation=SMOOTH[act])
    feat = state * (2 * order + 1)
    ocfg = O.ResMLPConfig(feat + emb, feat, tuple(widths), act)
    sd = {k: v.detach().double() for k, v in net.state_dict().items()}
    x = torch.randn(B, L, state)
    t = torch.rand(())

    def eps64(xx, tt):
        return O.mc_score_net(lambda a, b, c=None: O.score_net(sd, 'kernel.', ocfg, a, b, c), order, xx, tt)
    xo = x.double().requires_grad_(True)
    ref = eps64(xo, t.double())
    g = torch.randn(ref.shape, dtype=torch.float64)
    gref, = torch.autograd.grad(ref, xo, g)
    net = net.to(dev)
    xs = x.to(dev).requires_grad_(True)
    try:
        out = net(xs, t.to(dev))
        gout, = torch.autograd.grad(out, xs, g.float().to(dev))
    except Exception as e:  # noqa: BLE001
        return cfg, f'EXCEPTION {type(e).__name__}: {e}'
    return cfg, check('forward', out, ref, cfg) or check('vjp', gout, gref, cfg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=40)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    dev = torch.device('cuda:0')
    bad = 0
    for i in range(args.cases):
        fn = guided_case if i % 3 else local_case
        cfg, msg = fn(rng, dev, i + 7919 * args.seed)
        if msg:
            bad += 1
            print(f'FAIL case {i}: {msg}\n     {cfg}', flush=True)
    print(f'{args.cases - bad}/{args.cases} guided / local-path cases within 1e-4')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
