#!/usr/bin/env python3
"""Randomised parity sweep of conv_h2 (the OPT-IN f16 x 2 route of the block convolutions, csrc/conv_h2.hip) against torch float64.

    SDA_MULTIPLY=f16x2 python tests/fuzz/h2_fuzz.py [--cases 200] [--seed 0]

Every case draws channel counts (multiples of 96 up to 384: 1 .. 4 cout tiles -- and, round 6, multiples of 64 / 32: the 64-cout tile, K % 32), a NON-SQUARE image size in multiples of 16, a batch,
forward / backward-data packing, the padding, one of the loader fusions of the reference's blocks (none, SiLU, LayerNorm, modulation +
LayerNorm), one of the epilogues (none, x act'(z), + residual), bias on / off, an input magnitude between 1e-4 and 1e4 and where the
input scale comes from (an absmax pass, a loose static bound, the LayerNorm bound), runs the launch through the C ABI and compares
with a float64 convolution at 4e-6 of max |ref| (the kernel's error class; the fp32 Winograd kernel is run beside it).  The persistent
tile walk sees tile counts from 1 to ~300 (fewer / more than the 256 workgroups, not multiples of 8)."""
import argparse
import math
import os
import random
import sys

os.environ.setdefault('SDA_MULTIPLY', 'f16x2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sda_amd import ops  # noqa: E402
from sda_amd.engine import launch_conv, planar_source  # noqa: E402


def ref64(x, w, bias, circular, transpose, ln, mod, act_in, dact_z, res):
    x, w = x.double().cpu(), w.double().cpu()
    if ln:
        u = x + (0 if mod is None else mod.double().cpu().reshape(1, -1, 1, 1))
        var, mean = torch.var_mean(u, dim=1, unbiased=True, keepdim=True)
        x = (u - mean) / torch.sqrt(var + 1e-5)
    if act_in:
        x = F.silu(x)
    xp = F.pad(x, (1, 1, 1, 1), mode='circular') if circular else F.pad(x, (1, 1, 1, 1))
    if transpose:
        w = w.flip(2, 3).transpose(0, 1)
    y = F.conv2d(xp, w, None if bias is None else bias.double().cpu())
    if dact_z is not None:
        z = dact_z.double().cpu()
        s = torch.sigmoid(z)
        y = y * (s * (1 + z * (1 - s)))
    if res is not None:
        y = y + res.double().cpu()
    return y


def tail_case(rng, dev, idx, form, cin, cout, hs, ws, n, circular, tol):
    """One launch of the parity-class (up-sampled tail) or parity-plane (its pooled VJP) form against float64; returns 1 on failure."""
    w = (torch.rand(cout, cin, 3, 3, device=dev) * 2 - 1) / math.sqrt(cin * 9)
    scale = 10.0 ** rng.randint(-3, 3)
    if form == 'up':
        ln, with_res = rng.random() < 0.7, rng.random() < 0.5
        x = torch.randn(n, cin, hs, ws, device=dev) * scale
        b = torch.randn(cout, device=dev)
        pk = ops.PackedConv(w, b)
        out = torch.full((n, cout, 2 * hs, 2 * ws), float('nan'), device=dev)
        res = torch.randn_like(out) * scale if with_res else None
        kw = dict(circular=circular, bias=pk.bias, up=(2, 2), res=res)
        x64 = x.double().cpu()
        if ln:
            var, mean = torch.var_mean(x, dim=1, unbiased=True)
            kw['ln'] = (mean.reshape(-1).contiguous(), (1 / torch.sqrt(var + 1e-5)).reshape(-1).contiguous())
            v64, m64 = torch.var_mean(x64, dim=1, unbiased=True, keepdim=True)
            x64 = (x64 - m64) / torch.sqrt(v64 + 1e-5)
        xu = x64.repeat_interleave(2, -1).repeat_interleave(2, -2)
        xp = F.pad(xu, (1, 1, 1, 1), mode='circular') if circular else F.pad(xu, (1, 1, 1, 1))
        ref = F.conv2d(xp, w.double().cpu(), b.double().cpu())
        if res is not None:
            ref = ref + res.double().cpu()
        d = launch_conv(pk, planar_source(x), out, 2 * hs, 2 * ws, x_amax=None if ln else ops.absmax(x, pk.in_amax), **kw)
        ok = bool(d.w_h2) and d.up_h == 2
    elif form in ('s2', 'zins'):
        b = torch.randn(cout, device=dev)
        x = torch.randn(n, cin, 2 * hs, 2 * ws, device=dev) * scale
        x64 = x.double().cpu().requires_grad_(True)
        xp = F.pad(x64, (1, 1, 1, 1), mode='circular') if circular else F.pad(x64, (1, 1, 1, 1))
        y64 = F.conv2d(xp, w.double().cpu(), b.double().cpu() if form == 's2' else None, stride=2)
        if form == 's2':
            pk = ops.PackedConv(w, b)
            ref = y64.detach()
            out = torch.full((n, cout, hs, ws), float('nan'), device=dev)
            d = launch_conv(pk, planar_source(x), out, hs, ws, circular=circular, stride=(2, 2), bias=pk.bias, x_amax=ops.absmax(x, pk.in_amax))
            ok = bool(d.w_h2) and d.stride_h == 2
        else:
            pk = ops.PackedConv(w, None, transpose=True)
            g = torch.randn(n, cout, hs, ws, device=dev) * scale
            skip = torch.randn(n, cin, 2 * hs, 2 * ws, device=dev) * scale if rng.random() < 0.6 else None
            ref, = torch.autograd.grad(y64, x64, g.double().cpu())
            if skip is not None:
                ref = ref + skip.double().cpu()
            out = torch.full((n, cin, 2 * hs, 2 * ws), float('nan'), device=dev)
            d = launch_conv(pk, planar_source(g), out, 2 * hs, 2 * ws, circular=circular, zins=(2, 2), res=skip, x_amax=ops.absmax(g, pk.in_amax))
            ok = bool(d.w_h2) and d.zins_h == 2
    else:
        pk = ops.PackedConv(w, None, transpose=True)
        g = torch.randn(n, cout, 2 * hs, 2 * ws, device=dev) * scale
        x64 = torch.zeros(n, cin, hs, ws, dtype=torch.float64, requires_grad=True)
        xu = x64.repeat_interleave(2, -1).repeat_interleave(2, -2)
        xp = F.pad(xu, (1, 1, 1, 1), mode='circular') if circular else F.pad(xu, (1, 1, 1, 1))
        ref, = torch.autograd.grad(F.conv2d(xp, w.double().cpu()), x64, g.double().cpu())
        out = torch.full((n, cin, hs, ws), float('nan'), device=dev)
        d = launch_conv(pk, planar_source(g), out, 2 * hs, 2 * ws, circular=circular, pool=(2, 2), x_amax=ops.absmax(g, pk.in_amax))
        ok = d is not None and bool(d.w_h2) and d.pool_h == 2
    e = ((out.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    if not ok or not (e <= tol):
        print(f'FAIL case {idx}: {form} cin {cin} cout {cout} {hs}x{ws} n {n} {"circ" if circular else "zero"} x~{scale:.0e}: served {ok}, err {e:.2e}')
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--tol', type=float, default=4e-6)
    args = ap.parse_args()
    assert ops.MULTIPLY == 'f16x2', 'run with SDA_MULTIPLY=f16x2'
    dev = torch.device('cuda:0')
    rng = random.Random(args.seed)
    torch.manual_seed(args.seed)
    worst, worst32, fails, served = 0.0, 0.0, 0, 0
    for idx in range(args.cases):
        cin, cout = rng.choice([96, 96, 192, 288, 384, 64, 128, 256, 320]), rng.choice([96, 96, 192, 288, 384, 64, 128, 256, 320])
        h, w_ = 16 * rng.choice([1, 1, 2, 3, 4, 6]), 16 * rng.choice([1, 2, 2, 3, 5])
        n = rng.choice([1, 1, 2, 3, 5, 9])
        if n * cin * cout * h * w_ > 3.5e9:
            n = 1
        transpose, circular = rng.random() < 0.4, rng.random() < 0.6
        form = rng.choice(['conv', 'conv', 'conv', 'up', 'pool', 's2', 'zins'])   # up / pool: the tails and their VJP; s2 / zins: the stride-2 heads and theirs
        if form != 'conv':
            fails_here = tail_case(rng, dev, idx, form, cin, cout, h, w_, n, circular, args.tol)
            served += 1
            fails += fails_here
            continue
        loader = rng.choice(['plain', 'plain', 'silu', 'ln', 'modln'])
        epi = rng.choice(['none', 'none', 'res', 'dact']) if loader != 'ln' or True else 'none'
        scale = 10.0 ** rng.randint(-4, 4)
        cx = cout if transpose else cin
        co = cin if transpose else cout
        x = torch.randn(n, cx, h, w_, device=dev) * scale
        w = (torch.rand(cout, cin, 3, 3, device=dev) * 2 - 1) / math.sqrt(cin * 9)
        b = torch.randn(co, device=dev) if (not transpose and rng.random() < 0.7) else None
        pk = ops.PackedConv(w, b, transpose=transpose)
        assert pk.h2 is not None
        out = torch.empty(n, co, h, w_, device=dev)
        kw = dict(circular=circular, bias=pk.bias)
        mod = None
        if loader in ('ln', 'modln'):
            if loader == 'modln':
                mod = torch.randn(cx, device=dev) * scale
                kw.update(mod=mod, mod_sn=0)
            u = x + (0 if mod is None else mod.reshape(1, -1, 1, 1))
            var, mean = torch.var_mean(u, dim=1, unbiased=True)
            kw['ln'] = (mean.reshape(-1).contiguous(), (1 / torch.sqrt(var + 1e-5)).reshape(-1).contiguous())
            xa = None
        else:
            src = rng.choice(['pass', 'loose'])
            xa = ops.absmax(x, pk.in_amax) if src == 'pass' else torch.full((1,), float(x.abs().max()) * rng.choice([1.0, 3.0, 17.0]), device=dev)
        if loader == 'silu':
            kw['act_in'] = 1
        dz = rs = None
        if epi == 'dact':
            dz = torch.randn_like(out)
            kw.update(dact_z=dz, act_d=1)
        elif epi == 'res':
            rs = torch.randn_like(out) * scale
            kw['res'] = rs
        r = ref64(x, w, b, circular, transpose, loader in ('ln', 'modln'), mod, loader == 'silu', dz, rs)
        out.fill_(float('nan'))
        d = launch_conv(pk, planar_source(x), out, h, w_, x_amax=xa, out_amax=pk.out_amax, **kw)
        if not d.w_h2:
            print(f'case {idx}: NOT served by conv_h2: cin {cin} cout {cout} {h}x{w_}')
            continue
        served += 1
        e = ((out.double().cpu() - r).abs().max() / r.abs().max()).item()
        am_dev, am_true = pk.out_amax.item(), out.abs().max().item()
        keep = pk.h2
        pk.h2 = None
        out32 = torch.empty_like(out)
        launch_conv(pk, planar_source(x), out32, h, w_, **kw)
        pk.h2 = keep
        e32 = ((out32.double().cpu() - r).abs().max() / r.abs().max()).item()
        worst, worst32 = max(worst, e), max(worst32, e32)
        bad = not (e <= args.tol) or abs(am_dev - am_true) > 1e-6 * am_true
        if bad:
            fails += 1
            print(f'FAIL case {idx}: cin {cin} cout {cout} {h}x{w_} n {n} {"bwd" if transpose else "fwd"} {"circ" if circular else "zero"} {loader} {epi} '
                  f'bias {b is not None} x~{scale:.0e}: h2 {e:.2e} (fp32 kernel {e32:.2e}), out_amax {am_dev:.5g} / {am_true:.5g}')
    print(f'{args.cases} cases (seed {args.seed}): {served} served by conv_h2, {fails} failures; worst error vs float64 / max |ref|: h2 {worst:.2e}, '
          f'the fp32 kernels on the same launches {worst32:.2e}')
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
