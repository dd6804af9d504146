#!/usr/bin/env python3
"""Randomised parity sweep of whole score networks: ScoreUNet / MCScoreNet forward and input-VJP on the GPU against the
oracle evaluated in float64 (autograd through the oracle for the VJP).

    python tests/fuzz/net_fuzz.py [--cases 60] [--seed 0]

Random architectures (1-3 levels, 1-3 blocks, channel widths incl. the 96-multiples that take the Winograd kernel,
1-D / 2-D, zero / circular padding, any activation), random batch / window shapes, shared or per-sample times."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import sda_oracle as O  # noqa: E402
from sda_amd.score import MCScoreNet, ScoreUNet  # noqa: E402

ACTS = {'SiLU': nn.SiLU, 'GELU': nn.GELU, 'ELU': nn.ELU, 'ReLU': nn.ReLU, 'SELU': nn.SELU}


def _min_preact(run):
    """min |z| over every activation input of one oracle evaluation."""
    seen, orig = [], O.activation

    def recording(name):
        f = orig(name)

        def g(z):
            seen.append(z.detach().abs().min().item())
            return f(z)
        return g
    O.activation = recording
    try:
        with torch.no_grad():
            run()
    finally:
        O.activation = orig
    return min(seen) if seen else 1.0


SPATIAL3 = False            # --spatial3: every case is a 3-D U-Net (the general kernel of csrc/conv3d.hip)


def one_case(rng, dev, idx):
    spatial = rng.choice([1, 2, 2])
    if SPATIAL3:
        spatial = 3
    depth = rng.choice([1, 2, 2, 3])
    widths = {1: [4, 8, 24, 32, 64, 96], 2: [4, 8, 16, 32, 64, 96], 3: [4, 8, 32]}[depth]      # (32 / 64: the 32- and 64-cout Winograd tiles, round 6)
    c0 = rng.choice(widths)
    hidden = tuple(c0 * 2 ** i for i in range(depth))
    blocks = tuple(rng.choice([1, 2, 3]) for _ in range(depth))
    act = rng.choice(list(ACTS))
    pad = rng.choice(['zeros', 'circular'])
    mc = rng.random() < 0.5
    state = rng.choice([1, 2, 3])
    order = rng.choice([1, 2]) if mc else 0
    channels = state * (2 * order + 1) if mc else rng.choice([1, 2, 3, 5])
    context = rng.choice([0, 0, 1, 2]) if not mc else 0
    mult = 2 ** (depth - 1)
    size = [mult * rng.choice([1, 2, 3, 4, 8]) for _ in range(spatial)]
    if spatial == 2 and c0 >= 64:
        size = [min(s, 16) for s in size]
    if spatial == 3:
        size = [min(s, 8 if c0 < 64 else 4) for s in size]
        size = [max(s, mult) for s in size]
    emb = rng.choice([8, 16])
    cfg = dict(spatial=spatial, hidden=hidden, blocks=blocks, act=act, pad=pad, mc=mc, order=order, channels=channels,
               context=context, size=size)
    torch.manual_seed(9000 + idx)
    kw = dict(embedding=emb, hidden_channels=hidden, hidden_blocks=blocks, kernel_size=3, activation=ACTS[act],
              spatial=spatial, padding_mode=pad)
    ocfg = O.UNetConfig(channels + context, channels, emb, hidden, blocks, 3, 2, act, spatial, pad)
    B = rng.choice([1, 2, 3])
    per_sample_t = rng.random() < 0.4
    if mc:
        net = MCScoreNet(state, order=order, **kw)
        L = 2 * order + rng.choice([1, 2, 4])
        x = torch.randn(B, L, state, *size)
        t = torch.rand(B, L - 2 * order) if per_sample_t else torch.rand(())    # one time per window, or one for all
        c = None
    else:
        net = ScoreUNet(channels, context, **kw)
        x = torch.randn(B, channels, *size)
        t = torch.rand(B) if per_sample_t else torch.rand(())
        c = torch.randn(B, context, *size) if context else None
    cfg.update(B=B, per_sample_t=per_sample_t, x=tuple(x.shape))
    for p in net.parameters():                       # widen the default init so that every path carries signal
        p.data.mul_(1.5)
    sd = {k: v.detach().double() for k, v in net.state_dict().items()}

    def oracle(xx, tt, cc, s=sd):
        if mc:
            kern = lambda a, b, _c=None: O.score_unet(s, 'kernel.', ocfg, a, b, None)
            return O.mc_score_net(kern, order, xx, tt)
        return O.score_unet(s, '', ocfg, xx, tt, cc)

    xo = x.double().requires_grad_(True)
    ref = oracle(xo, t.double(), None if c is None else c.double())
    g = torch.randn(ref.shape, dtype=torch.float64)
    gref, = torch.autograd.grad(ref, xo, g)
    # ReLU / SELU have a discontinuous derivative: a pre-activation within fp32 round-off of 0 flips act'(z) between any
    # two fp32 evaluations, and the flip spreads over the whole receptive field of the VJP.  Such cases say nothing about
    # the kernels: skip them (the oracle's own pre-activations tell).
    if act in ('ReLU', 'SELU') and _min_preact(lambda: oracle(x.double(), t.double(), None if c is None else c.double())) < 2e-5:
        return cfg, 'SKIP'
    net = net.to(dev)
    xs = x.to(dev).requires_grad_(True)
    try:
        out = net(xs, t.to(dev)) if c is None else net(xs, t.to(dev), c.to(dev))
        gout, = torch.autograd.grad(out, xs, g.float().to(dev))
    except Exception as e:  # noqa: BLE001
        return cfg, f'EXCEPTION {type(e).__name__}: {e}'
    for name, got, want in (('forward', out, ref), ('vjp', gout, gref)):
        got = got.detach().cpu().double()
        if not torch.isfinite(got).all():
            return cfg, f'{name}: non-finite output'
        scale = want.abs().max().item() + 1e-30
        err = (got - want).abs().max().item()
        if err > 1e-4 * scale:
            nbad = int(((got - want).abs() > 1e-4 * scale).sum())
            return cfg, f'{name}: max abs err {err:.3e} vs scale {scale:.3e} ({nbad} of {got.numel()} elements off)'
    return cfg, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=60)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--only', type=int, default=-1, help='report only this case index (the sequence is still drawn)')
    ap.add_argument('--spatial3', action='store_true', help='3-D U-Nets only (small volumes)')
    args = ap.parse_args()
    global SPATIAL3
    SPATIAL3 = args.spatial3
    rng = random.Random(args.seed)
    dev = torch.device('cuda:0')
    bad = skipped = 0
    for i in range(args.cases):
        cfg, msg = one_case(rng, dev, i + 7919 * args.seed)
        if msg == 'SKIP':
            skipped += 1
            continue
        if msg:
            bad += 1
            print(f'FAIL case {i}: {msg}\n     {cfg}', flush=True)
    print(f'{args.cases - bad - skipped}/{args.cases - skipped} networks within 1e-4 (forward and VJP); '
          f'{skipped} ill-conditioned cases skipped (a ReLU/SELU pre-activation within 2e-5 of its kink)')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
