#!/usr/bin/env python3
"""conv_h2 (f16 x 2 emulation of the fp32 multiply, csrc/conv_h2.hip) against float64 and against the fp32 kernels, per layer type.

    SDA_MULTIPLY=f16x2 python tools/h2_check.py [--quick]

Correctness: small batches, every loader / epilogue the reference's blocks use, forward and backward-data packings, circular and zero
padding -- max |err| / max |ref| against a float64 CPU convolution, next to the same launch on the fp32 kernels (conv_wino4 / direct).
Timing: the configs[3] / configs[2] layer shapes, HIP events, 20 warm-up launches, fp32 kernel vs h2 kernel on the same operands."""
import math
import os
import sys

os.environ.setdefault('SDA_MULTIPLY', 'f16x2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from sda_amd import ops
from sda_amd.engine import launch_conv, planar_source

assert ops.MULTIPLY == 'f16x2'
dev = torch.device('cuda:0')
quick = '--quick' in sys.argv
timing_only = '--timing-only' in sys.argv
plain_only = '--plain' in sys.argv


def ref64(x, w, bias, circular, transpose, ln, mod, act_in, dact_z, res):
    x = x.double().cpu()
    w = w.double().cpu()
    if ln is not None:
        u = x + (0 if mod is None else mod.double().cpu().reshape(1, -1, 1, 1))
        var, mean = torch.var_mean(u, dim=1, unbiased=True, keepdim=True)
        x = (u - mean) / torch.sqrt(var + 1e-5)
    if act_in:
        x = F.silu(x)
    xp = F.pad(x, (1, 1, 1, 1), mode='circular') if circular else F.pad(x, (1, 1, 1, 1))
    if transpose:                                   # backward-data of conv(w): conv with flipped taps, cin <-> cout
        w = w.flip(2, 3).transpose(0, 1)
    y = F.conv2d(xp, w, None if bias is None else bias.double().cpu())
    if dact_z is not None:
        z = dact_z.double().cpu()
        s = torch.sigmoid(z)
        y = y * (s * (1 + z * (1 - s)))
    if res is not None:
        y = y + res.double().cpu()
    return y


def run(pk, x, out, *, circular, ln=None, mod=None, act_in=0, dact_z=None, res=None, h2=True, xa='pass'):
    kw = dict(circular=circular, bias=pk.bias, act_in=act_in, res=res)
    if ln is not None:
        kw['ln'] = ln
        if mod is not None:
            kw.update(mod=mod, mod_sn=0)
    if dact_z is not None:
        kw.update(dact_z=dact_z, act_d=1)
    keep = pk.h2
    if not h2:
        pk.h2 = None
    try:
        if xa == 'pass':
            xa = None if ln is not None or not h2 else ops.absmax(x, pk.in_amax)
        d = launch_conv(pk, planar_source(x), out, out.shape[2], out.shape[3], x_amax=xa, out_amax=pk.out_amax if h2 else None, **kw)
    finally:
        pk.h2 = keep
    return d


def stats(x, mod):
    u = x + (0 if mod is None else mod.reshape(1, -1, 1, 1))
    var, mean = torch.var_mean(u, dim=1, unbiased=True)
    return mean.reshape(-1).contiguous(), (1 / torch.sqrt(var + 1e-5)).reshape(-1).contiguous()


torch.manual_seed(0)
print('--- correctness: max |err| / max |ref| against float64')
worst = 0.0
cases = [('plain', {}), ('mod+LN', dict(ln=True, mod=True)), ('LN', dict(ln=True)), ('SiLU+res', dict(act_in=1, res=True)), ('x act\'(z)', dict(dact=True))]
for cin, cout, hw, n in (() if timing_only else ((96, 96, 32, 2), (192, 96, 16, 3), (96, 192, 48, 1), (384, 384, 16, 2))):
    for transpose in (False, True):
        for circular in (True, False):
            for name, fz in cases:
                if quick and (transpose or not circular) and name not in ('plain', "x act'(z)"):
                    continue
                scale = 10.0 ** torch.randint(-3, 4, (1,)).item()           # the input's magnitude must not matter
                x = torch.randn(n, cout if transpose else cin, hw, hw, device=dev) * scale
                w = (torch.rand(cout, cin, 3, 3, device=dev) * 2 - 1) / math.sqrt(cin * 9)
                b = None if transpose else torch.randn(cout, device=dev)
                pk = ops.PackedConv(w, b, transpose=transpose)
                assert pk.h2 is not None
                co = cin if transpose else cout
                out = torch.empty(n, co, hw, hw, device=dev)
                mod = torch.randn(x.shape[1], device=dev) * scale if fz.get('mod') else None
                ln = stats(x, mod) if fz.get('ln') else None
                dz = torch.randn_like(out) if fz.get('dact') else None
                rs = torch.randn_like(out) * scale if fz.get('res') else None
                kw = dict(circular=circular, ln=ln, mod=mod, act_in=fz.get('act_in', 0), dact_z=dz, res=rs)
                r = ref64(x, w, b, circular, transpose, ln, mod, kw['act_in'], dz, rs)
                d = run(pk, x, out, **kw)
                assert d.w_h2, 'the h2 kernel did not serve the launch'
                e_h2 = ((out.double().cpu() - r).abs().max() / r.abs().max()).item()
                amax_dev, amax_true = pk.out_amax.item(), out.abs().max().item()
                out32 = torch.empty_like(out)
                run(pk, x, out32, h2=False, **kw)
                e_32 = ((out32.double().cpu() - r).abs().max() / r.abs().max()).item()
                worst = max(worst, e_h2)
                flag = '' if e_h2 < 2e-6 and abs(amax_dev - amax_true) <= 1e-6 * amax_true else '   <-- CHECK'
                print(f'{cin:3d}->{cout:3d} @{hw:3d} n={n} {"bwd" if transpose else "fwd"} {"circ" if circular else "zero"} {name:10s} x~{scale:7.0e}  '
                      f'h2 {e_h2:.2e}   fp32 kernel {e_32:.2e}   out_amax {amax_dev:.4g} / {amax_true:.4g}{flag}')
print(f'worst h2 error {worst:.2e}')

print('--- timing (ms per launch; fp32 kernel -> h2 kernel)')
shapes = [('96->96 @256^2 x30', 96, 96, 256, 30), ('192->192 @128^2 x60', 192, 192, 128, 60), ('384->384 @64^2 x120', 384, 384, 64, 120),
          ('96->96 @64^2 x896', 96, 96, 64, 896)]
if quick:
    shapes = shapes[:1] + shapes[2:3]
for label, cin, cout, hw, n in shapes:
    for name, fz in (cases[:1] if plain_only else cases):
        x = torch.randn(n, cin, hw, hw, device=dev)
        w = (torch.rand(cout, cin, 3, 3, device=dev) * 2 - 1) / math.sqrt(cin * 9)
        pk = ops.PackedConv(w, torch.randn(cout, device=dev))
        out = torch.empty(n, cout, hw, hw, device=dev)
        mod = torch.randn(cin, device=dev) if fz.get('mod') else None
        ln = stats(x, mod) if fz.get('ln') else None
        dz = torch.randn_like(out) if fz.get('dact') else None
        rs = torch.randn_like(out) if fz.get('res') else None
        kw = dict(circular=True, ln=ln, mod=mod, act_in=fz.get('act_in', 0), dact_z=dz, res=rs)
        ms = {}
        xa0 = None if ln is not None else ops.absmax(x, torch.zeros(1, device=dev))
        for mode in ('f32', 'h2', 'h2+pass'):
            h2 = mode != 'f32'
            xa = 'pass' if mode == 'h2+pass' else xa0
            for _ in range(20):
                run(pk, x, out, h2=h2, xa=xa, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(pk, x, out, h2=h2, xa=xa, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms[mode] = e0.elapsed_time(e1) / 10
        fl = 2.0 * n * hw * hw * cout * cin * 9
        print(f'{label:22s} {name:10s} fp32 {ms["f32"]:7.3f}  h2 {ms["h2"]:7.3f} ms ({ms["f32"] / ms["h2"]:.2f}x)  with an absmax pass over x {ms["h2+pass"]:7.3f} '
              f'({ms["f32"] / ms["h2+pass"]:.2f}x)   h2: {fl / ms["h2"] / 1e9:.0f} TFLOP/s direct-equivalent = {3 * fl / ms["h2"] / 1e9 / 2500:.3f} of the f16 peak at 3 products')
