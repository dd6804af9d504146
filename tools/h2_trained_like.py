#!/usr/bin/env python3
"""The opt-in f16 x 2 multiply (csrc/conv_h2.hip) on operands shaped like a TRAINED checkpoint's: weights whose
output (and input) channels carry scales spread log-uniformly over 1e-3 .. 1e1 -- a per-tensor power-of-two scale then leaves the small
channels' `lo` halves in f16's subnormal range -- and activations with a 1 % heavy tail (cubed Gaussians: max |x| ~ 1e3 x the typical
value).  Error against float64, of the f16 x 2 kernel and of the fp32 kernels on the same launch: over the whole tensor (/ max |ref|,
the suite's criterion) and PER OUTPUT CHANNEL (/ that channel's max |ref|: what a per-tensor scale could hide).

    python tools/h2_trained_like.py            (GPU box)"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sda_amd import ops  # noqa: E402
from sda_amd.engine import launch_conv, planar_source  # noqa: E402

dev = torch.device('cuda:0')


def operands(cin, cout, hw, n, seed, tail=True, chan=True):
    g = torch.Generator().manual_seed(seed)
    w = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / math.sqrt(cin * 9)
    if chan:
        so = 10.0 ** (torch.rand(cout, generator=g) * 4 - 3)             # per-output-channel scale, log-uniform in 1e-3 .. 1e1
        si = 10.0 ** (torch.rand(cin, generator=g) * 2 - 1)              # per-input-channel scale, 1e-1 .. 1e1
        w = w * so[:, None, None, None] * si[None, :, None, None]
    x = torch.randn(n, cin, hw, hw, generator=g)
    if tail:
        m = torch.rand(x.shape, generator=g) < 0.01
        x = torch.where(m, (3 * torch.randn(x.shape, generator=g)) ** 3, x)
    return x, w


def errors(out, ref):
    d = (out.double().cpu() - ref).abs()
    whole = (d.max() / ref.abs().max()).item()
    per = (d.amax(dim=(0, 2, 3)) / ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-300))
    return whole, per.max().item(), per.median().item()


def run(cin, cout, hw, n, seed, **kw):
    x, w = operands(cin, cout, hw, n, seed, **kw)
    ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode='circular'), w.double())
    res = {}
    for mult in ('f32', 'f16x2'):
        prev = ops.set_multiply(mult)
        try:
            xd, wd = x.to(dev), w.to(dev)
            pk = ops.PackedConv(wd, None)
            out = torch.empty(n, cout, hw, hw, device=dev)
            xa = ops.absmax(xd, pk.in_amax) if pk.h2 is not None else None
            d = launch_conv(pk, planar_source(xd), out, hw, hw, circular=True, x_amax=xa)
            torch.cuda.synchronize()
            res[mult] = (bool(d.w_h2),) + errors(out, ref)
        finally:
            ops.set_multiply(prev)
    return res


if __name__ == '__main__':
    print('case | kernel | whole-tensor err / max|ref| | worst per-output-channel err / that channel\'s max|ref| | median per channel')
    for name, kw in (('uniform weights, Gaussian activations', dict(tail=False, chan=False)), ('heavy-tailed activations (1 % cubed)', dict(chan=False)),
                     ('channel-scaled weights (1e-3..1e1 out, 1e-1..1e1 in)', dict(tail=False)), ('both', {})):
        for cin, cout, hw, n in ((96, 96, 32, 2), (192, 192, 32, 2), (384, 384, 16, 2)):
            r = run(cin, cout, hw, n, 11 + cin, **kw)
            for mult, (h2, whole, worst, med) in r.items():
                print(f'{name:55s} {cin:3d}->{cout:3d} @{hw} | {mult:5s} (h2 served: {int(h2)}) | {whole:.2e} | {worst:.2e} | {med:.2e}')
