#!/usr/bin/env python3
"""Split-bf16 emulation of the fp32 multiply in the Winograd F(2x2,3x3) convolution, against float64 (CPU, torch): the numerics half
of VERDICT r2 item 8 (needs sign-off; nothing in the product uses this).  A value x is split x = hi + mid + lo with hi = bf16(x),
mid = bf16(x - hi), lo = bf16(x - hi - mid) (each difference exact in fp32); a product of two bf16 values is exact in fp32, so an MFMA
chain v_mfma_f32_16x16x32_bf16 over the pieces with fp32 accumulation computes
    3 products:  hi.hi' + hi.mid' + mid.hi'                               (drops terms of relative size 2^-16)
    6 products:  ... + mid.mid' + hi.lo' + lo.hi'                          (drops 2^-24 .. 2^-25)
Here: U = G g G^T and V = B^T d B are formed in fp32 (as conv_wino4 does), split, multiplied piecewise with fp32 accumulation over the
channels, and the inverse transform runs in fp32.  Compared with the all-fp32 Winograd form and the direct fp32 convolution.

    python tools/bf16_split_numerics.py
"""
import math

import torch

torch.manual_seed(0)
BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = bf(r)
        parts.append(p)
        r = r - p
    return parts


def wino(x, w, pairs):
    """pairs: None = plain fp32 products; else list of (i, j) piece index pairs to multiply"""
    N, C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1), mode='circular')
    pat = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # N, C, th, tw, 4, 4
    V = torch.einsum('ij,nctujk,lk->nctuil', BT, pat, BT)
    U = torch.einsum('ij,kcjl,ml->kcim', G, w, G)
    if pairs is None:
        M = torch.einsum('kcil,nctuil->nktuil', U, V)
    else:
        npieces = 1 + max(max(p) for p in pairs)
        Us, Vs = split(U, npieces), split(V, npieces)
        M = None
        for i, j in pairs:                                                      # fp32 accumulation across the piece products
            t = torch.einsum('kcil,nctuil->nktuil', Us[i], Vs[j])
            M = t if M is None else M + t
    Y = torch.einsum('ij,nktujl,ml->nktuim', AT, M, AT)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], H, W)


def ref(x, w):
    return torch.nn.functional.conv2d(torch.nn.functional.pad(x.double(), (1, 1, 1, 1), mode='circular'), w.double())


def rel(a, b):
    return ((a.double() - b).abs().max() / b.abs().max()).item()


def rms(a, b):
    return ((a.double() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


P3 = [(0, 0), (0, 1), (1, 0)]
P4 = P3 + [(1, 1)]
P6 = P4 + [(0, 2), (2, 0)]
print('relative error against float64 (max / rms over the output), unit-variance activations, torch default conv init')
for C, K, S in ((96, 96, 32), (192, 192, 16), (384, 384, 16)):
    x = torch.randn(2, C, S, S)
    w = (torch.rand(K, C, 3, 3) * 2 - 1) / math.sqrt(C * 9)
    r = ref(x, w)
    d32 = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode='circular'), w)
    rows = [('direct fp32', d32), ('Winograd fp32 (the product kernel)', wino(x, w, None)), ('Winograd split-bf16, 3 products', wino(x, w, P3)),
            ('Winograd split-bf16, 4 products', wino(x, w, P4)), ('Winograd split-bf16, 6 products', wino(x, w, P6)),
            ('Winograd plain bf16 (1 product)', wino(x, w, [(0, 0)]))]
    print(f'--- {C} -> {K} channels, {S} x {S}')
    for name, y in rows:
        print(f'   {name:40s} max {rel(y, r):.2e}   rms {rms(y, r):.2e}')
