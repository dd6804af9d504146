#!/usr/bin/env python3
"""Split-FP16 emulation of the fp32 multiply in the Winograd F(2x2,3x3) convolution, against float64 (CPU, torch) -- numerics only,
the follow-up to tools/bf16_split_numerics.py (nothing in the product uses this).

Why fp16 instead of bf16: fp16 carries 11 significand bits, so TWO pieces x = hi + lo (hi = fp16(x), lo = fp16(x - hi), the difference
exact in fp32) hold 22 of fp32's 24 bits -- in 4 bytes per value, the SAME operand bytes as fp32 (three bf16 pieces are 6 bytes: the LDS
feed is what killed that variant, profiles/r04_bf16x6_prototype.txt) -- and 3 products (hi.hi' + hi.lo' + lo.hi', dropping 2^-22) or 4
(+ lo.lo') on v_mfma_f32_16x16x32_f16 replace 8 v_mfma_f32_16x16x4_f32 per 32 channels.  fp16's narrow exponent is handled by one
power-of-two scale per tensor (max |value| -> 2^12: products of two such values stay far inside fp32, small elements' low pieces fall
into fp16 subnormals whose ABSOLUTE spacing, 2^-24, is 2^-36 of the tensor scale).

    python tools/f16_split_numerics.py
"""
import math

import torch

torch.manual_seed(0)
BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def split16(x):
    s = 2.0 ** (12 - math.ceil(math.log2(x.abs().max().item())))          # one power-of-two scale per tensor
    xs = x * s
    hi = xs.to(torch.float16).to(torch.float32)
    lo = (xs - hi).to(torch.float16).to(torch.float32)
    return (hi, lo), s


def wino(x, w, pairs):
    N, C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1), mode='circular')
    pat = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum('ij,nctujk,lk->nctuil', BT, pat, BT)
    U = torch.einsum('ij,kcjl,ml->kcim', G, w, G)
    if pairs is None:
        M = torch.einsum('kcil,nctuil->nktuil', U, V)
    else:
        Us, su = split16(U)
        Vs, sv = split16(V)
        M = None
        for i, j in pairs:                                                  # fp32 accumulation across the piece products
            t = torch.einsum('kcil,nctuil->nktuil', Us[i], Vs[j])
            M = t if M is None else M + t
        M = M * (1.0 / (su * sv))
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
print('relative error against float64 (max / rms over the output), torch default conv init')
for C, K, S, kind in ((96, 96, 32, 'unit-variance'), (192, 192, 16, 'unit-variance'), (384, 384, 16, 'unit-variance'),
                      (96, 96, 32, 'heavy-tailed (x^3)')):
    x = torch.randn(2, C, S, S)
    if kind != 'unit-variance':
        x = x ** 3                                                          # dynamic range 1e6 within the tensor
    w = (torch.rand(K, C, 3, 3) * 2 - 1) / math.sqrt(C * 9)
    r = ref(x, w)
    rows = [('Winograd fp32 (the product kernel)', wino(x, w, None)), ('Winograd split-fp16 x2, 3 products', wino(x, w, P3)),
            ('Winograd split-fp16 x2, 4 products', wino(x, w, P4)), ('Winograd plain fp16 (1 product)', wino(x, w, [(0, 0)]))]
    print(f'--- {C} -> {K} channels, {S} x {S}, {kind} activations')
    for name, y in rows:
        print(f'   {name:40s} max {rel(y, r):.2e}   rms {rms(y, r):.2e}')
