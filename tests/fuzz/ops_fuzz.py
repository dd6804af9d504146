#!/usr/bin/env python3
"""Randomised checks of the small HIP operators around the U-Net: observation operators (value vs the reference's
indexing / torch restatement, adjoint vs autograd and the dot-product identity <A x, r> = <x, A^T r>), fold / unfold
adjoints, and the predictor-corrector / guidance elementwise kernels, on random shapes.

    python tests/fuzz/ops_fuzz.py [--cases 200] [--seed 0]
"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from oracle import sda_oracle as O  # noqa: E402
from sda_amd import observe as Ob  # noqa: E402
from sda_amd import ops  # noqa: E402


def close(got, want, tol=1e-5):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    if got.shape != want.shape:
        return f'shape {tuple(got.shape)} vs {tuple(want.shape)}'
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    return None if err <= tol * scale else f'max abs err {err:.3e} vs scale {scale:.3e}'


def observe_case(rng, dev, idx):
    torch.manual_seed(100 + idx)
    kind = rng.choice(['subsample', 'subsample', 'coarsen', 'vorticity', 'compose'])
    cfg = dict(kind=kind)
    if kind == 'subsample':
        nd = rng.choice([1, 2, 3])
        lead = [rng.choice([1, 2, 3]) for _ in range(rng.choice([1, 2]))]
        tail = [rng.choice([1, 3, 8, 17, 32]) for _ in range(nd)]
        sl = []
        for n in tail:
            start = rng.randrange(0, n)
            step = rng.choice([1, 2, 3, 8])
            stop = rng.choice([None, None, rng.randrange(start + 1, n + 1)])
            sl.append(slice(start, stop, step))
        x = torch.randn(*lead, *tail)
        A = Ob.Subsample(sl)
        ref = lambda v: v[(Ellipsis,) + tuple(sl)]
        cfg.update(shape=tuple(x.shape), slices=[(s.start, s.stop, s.step) for s in sl])
    elif kind == 'coarsen':
        f = rng.choice([2, 4])
        x = torch.randn(rng.choice([1, 3]), rng.choice([1, 2, 5]), 2, f * rng.choice([1, 3, 8]), f * rng.choice([2, 4, 8]))
        A = Ob.Coarsen(f)
        ref = lambda v: O.coarsen(v, f)
        cfg.update(shape=tuple(x.shape), f=f)
    elif kind == 'vorticity':
        x = torch.randn(rng.choice([1, 2]), rng.choice([1, 4]), 2, rng.choice([4, 8, 18]), rng.choice([4, 16, 30]))
        A = Ob.Vorticity()
        ref = O.vorticity
        cfg.update(shape=tuple(x.shape))
    else:
        f = 2
        x = torch.randn(2, rng.choice([2, 6]), 2, 16, 32)
        A = Ob.Compose(Ob.Coarsen(f), Ob.Subsample((slice(None, None, 2), slice(None), slice(None), slice(None))))
        ref = lambda v: O.coarsen(v, f)[..., ::2, :, :, :]
        cfg.update(shape=tuple(x.shape))
    xr = x.double().requires_grad_(True)
    want = ref(xr)
    r = torch.randn(want.shape)
    gwant, = torch.autograd.grad(want, xr, r.double())
    try:
        got = A(x.to(dev))
        gx = A.adjoint(r.to(dev).contiguous(), tuple(x.shape))
    except Exception as e:  # noqa: BLE001
        return cfg, f'EXCEPTION {type(e).__name__}: {e}'
    msg = close(got, want) or close(gx, gwant)
    if msg:
        return cfg, msg
    lhs = (got.cpu().double() * r.double()).sum().item()
    rhs = (x.double() * gx.cpu().double()).sum().item()
    if abs(lhs - rhs) > 1e-4 * (abs(lhs) + abs(rhs) + 1e-9):
        return cfg, f'adjoint identity {lhs} vs {rhs}'
    return cfg, None


def fold_case(rng, dev, idx):
    torch.manual_seed(300 + idx)
    k = rng.choice([1, 2, 4])
    B, nw, C = rng.choice([1, 2, 5]), rng.choice([1, 2, 7, 30]), rng.choice([1, 2, 3, 40])
    rest = rng.choice([(1,), (5,), (4, 6), (16, 16)])
    hw = 1
    for r_ in rest:
        hw *= r_
    wl = 2 * k + 1
    cfg = dict(kind='fold', k=k, B=B, nw=nw, C=C, rest=rest)
    s = torch.randn(B, nw, wl * C, *rest)
    sr = s.double().requires_grad_(True)
    want = O.fold(sr, k)
    g = torch.randn(want.shape)
    gs_want, = torch.autograd.grad(want, sr, g.double())
    xr = torch.randn(B, nw + 2 * k, C, *rest).double().requires_grad_(True)
    u = O.unfold(xr, k)
    gw = torch.randn(u.shape)
    gx_want, = torch.autograd.grad(u, xr, gw.double())
    try:
        out = torch.empty(B, nw + 2 * k, C, *rest, device=dev)
        ops.fold(s.to(dev).contiguous(), B, nw, k, C, hw, out)
        g_s = torch.empty(B, nw, wl * C, *rest, device=dev)
        ops.fold_adjoint(g.to(dev).contiguous(), B, nw, k, C, hw, g_s)
        g_x = torch.empty(B, nw + 2 * k, C, *rest, device=dev)
        ops.unfold_adjoint(gw.to(dev).contiguous(), B, nw, k, C, hw, wl * C, g_x)
    except Exception as e:  # noqa: BLE001
        return cfg, f'EXCEPTION {type(e).__name__}: {e}'
    return cfg, close(out, want, 1e-7) or close(g_s, gs_want, 1e-7) or close(g_x, gx_want, 1e-6)


def pc_case(rng, dev, idx):
    torch.manual_seed(500 + idx)
    b = rng.choice([1, 2, 7])
    per = rng.choice([3, 64, 1000, 4097, 65536 + 5])
    cfg = dict(kind='pc', b=b, per=per)
    x, eps, z = torch.randn(b, per), torch.randn(b, per) * 0.7, torch.randn(b, per)
    r, c1, tau, sigma, mu = 0.93, -0.21, rng.choice([0.25, 0.5, 1.0]), 0.6, 0.8
    try:
        xd = x.to(dev).clone()
        ops.pc_predict(xd, eps.to(dev), r, c1)
        msg = close(xd, r * x + c1 * eps, 1e-6)
        if msg:
            return cfg, 'pc_predict: ' + msg
        xd = x.to(dev).clone()
        partial = torch.empty(b, ops.SUMSQ_CHUNKS, device=dev)
        ops.sumsq_partial(eps.to(dev), b, partial)
        ops.pc_correct(xd, eps.to(dev), z.to(dev), b, partial, tau, sigma)
        delta = tau / eps.double().square().mean(1, keepdim=True)
        want = x.double() - (delta * eps.double() + torch.sqrt(2 * delta) * z.double()) * sigma
        msg = close(xd, want, 2e-6)
        if msg:
            return cfg, 'pc_correct: ' + msg
        xhat = torch.empty(b, per, device=dev)
        ops.denoise(x.to(dev), eps.to(dev), mu, sigma, xhat)
        msg = close(xhat, (x.double() - sigma * eps.double()) / mu, 1e-6)
        if msg:
            return cfg, 'denoise: ' + msg
        out = torch.empty(b, per, device=dev)
        ops.guided_combine(eps.to(dev), z.to(dev), x.to(dev), mu, sigma, out)
        want = eps.double() - sigma * (z.double() / mu - (sigma / mu) * x.double())
        msg = close(out, want, 1e-6)
        if msg:
            return cfg, 'guided_combine: ' + msg
    except Exception as e:  # noqa: BLE001
        return cfg, f'EXCEPTION {type(e).__name__}: {e}'
    return cfg, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    dev = torch.device('cuda:0')
    bad = 0
    for i in range(args.cases):
        fn = (observe_case, fold_case, pc_case)[i % 3]
        cfg, msg = fn(rng, dev, i + 7919 * args.seed)
        if msg:
            bad += 1
            print(f'FAIL case {i}: {msg}\n     {cfg}', flush=True)
    print(f'{args.cases - bad}/{args.cases} operator cases pass')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
