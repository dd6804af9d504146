#!/usr/bin/env python3
"""conv_h2 on zeros vs random data (same launches): how much of a layer's time is the power-managed clock (tools/h2_power_probe.hip)
and how much is the kernel's own cycles.  SDA_MULTIPLY=f16x2 python tools/h2_zero_probe.py"""
import math, os, sys
os.environ.setdefault('SDA_MULTIPLY', 'f16x2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sda_amd import ops
from sda_amd.engine import launch_conv, planar_source
dev = torch.device('cuda:0')
for label, c, hw, n in (('96 @256^2 x30', 96, 256, 30), ('192 @128^2 x60', 192, 128, 60), ('384 @64^2 x120', 384, 64, 120)):
    for data in (('x = w = 0',) if '--zeros' in sys.argv else ('random', 'x = 0', 'x = w = 0')):
        x = torch.randn(n, c, hw, hw, device=dev) if data == 'random' else torch.zeros(n, c, hw, hw, device=dev)
        w = (torch.rand(c, c, 3, 3, device=dev) * 2 - 1) / math.sqrt(c * 9)
        if data == 'x = w = 0':
            w = w * 0 + 1e-30
        for epi in ('plain', '+res'):
            pk = ops.PackedConv(w, torch.randn(c, device=dev))
            out = torch.empty(n, c, hw, hw, device=dev)
            res = (torch.randn_like(out) if data == 'random' else torch.zeros_like(out)) if epi == '+res' else None
            xa = torch.ones(1, device=dev)
            def go():
                launch_conv(pk, planar_source(x), out, hw, hw, circular=True, bias=pk.bias, res=res, x_amax=xa, out_amax=None)
            for _ in range(20): go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): go()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            fl = 2.0 * n * hw * hw * c * c * 9
            print(f'{label:16s} {data:10s} {epi:6s} {ms:7.3f} ms   {3 * fl / ms / 1e9 / 2500:.3f} of the nominal f16 peak at 3 products')
