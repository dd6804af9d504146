#!/usr/bin/env python3
"""Quick A/B timing of conv_wino4 on four layer types at 64^2 (HIP events, 30 warm-up launches): for comparing tooling builds
selected through SDA_HIP_LIB.      SDA_HIP_LIB=... python tools/w4_quick_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sda_amd import ops
from sda_amd.engine import launch_conv, planar_source
dev = torch.device('cuda:0')
res = []
CASES = (('plain96', 96, 96, 64, 896, {}), ('modLN96', 96, 96, 64, 896, dict(ln=True, mod=True)), ('silu+res96', 96, 96, 64, 896, dict(silu=True, res=True)),
                                  ('dact96', 96, 96, 64, 896, dict(dact=True)), ('silu+res192', 192, 192, 32, 896, dict(silu=True, res=True)), ('dact192', 192, 192, 32, 896, dict(dact=True)),
                                  ('plain384', 384, 384, 16, 896, {}), ('silu+res384', 384, 384, 16, 896, dict(silu=True, res=True)), ('dact384', 384, 384, 16, 896, dict(dact=True)),
                                  ('uptail192', 192, 96, 64, 896, dict(ln=True, res=True, up=True)), ('pooled96', 96, 192, 64, 896, dict(pool=True)))
if os.environ.get('W4Q_BM64'):          # the 64-cout tile: the reference's default widths (64, 128, 256)
    CASES = (('plain64', 64, 64, 64, 960, {}), ('modLN64', 64, 64, 64, 960, dict(ln=True, mod=True)), ('silu+res64', 64, 64, 64, 960, dict(silu=True, res=True)),
             ('dact64', 64, 64, 64, 960, dict(dact=True)), ('plain128', 128, 128, 32, 960, {}), ('modLN128', 128, 128, 32, 960, dict(ln=True, mod=True)),
             ('silu+res128', 128, 128, 32, 960, dict(silu=True, res=True)), ('dact128', 128, 128, 32, 960, dict(dact=True)),
             ('plain256', 256, 256, 16, 960, {}), ('silu+res256', 256, 256, 16, 960, dict(silu=True, res=True)), ('dact256', 256, 256, 16, 960, dict(dact=True)),
             ('uptail128', 128, 64, 64, 960, dict(ln=True, res=True, up=True)), ('pooled64', 64, 128, 64, 960, dict(pool=True)))
for name, cin, cout, h, n, fz in CASES:
    hs = h // 2 if fz.get('up') else h
    x = torch.randn(n, cin, hs, hs, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    pk = ops.PackedConv(w, torch.randn(cout, device=dev))
    out = torch.empty(n, cout, h // 2, h // 2, device=dev) if fz.get('pool') else torch.empty(n, cout, h, h, device=dev)
    kw = dict(circular=True, bias=pk.bias)
    if fz.get('ln'): kw['ln'] = (torch.zeros(n * hs * hs, device=dev), torch.ones(n * hs * hs, device=dev))
    if fz.get('up'): kw['up'] = (2, 2)
    if fz.get('pool'): kw['pool'] = (2, 2)
    if fz.get('mod'): kw['mod'] = torch.randn(1, cin, device=dev)
    if fz.get('silu'): kw['act_in'] = 1
    if fz.get('res'): kw['res'] = torch.randn_like(out)
    if fz.get('dact'): kw.update(dact_z=torch.randn_like(out), act_d=1)
    for _ in range(30): launch_conv(pk, planar_source(x), out, h, h, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): launch_conv(pk, planar_source(x), out, h, h, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    issued = 4.0 if (fz.get('up') or fz.get('pool')) else 2.25          # zero-position kernels issue 54 of 96 MFMAs per stage
    res.append(f'{name} {ms:.3f} ms ({2.0 * n * h * h * cout * cin * 9 / ms / 1e9 / issued / 157.3:.3f})')
print(os.environ.get('SDA_HIP_LIB', 'product'), ' | '.join(res))
