#!/usr/bin/env python3
"""HIP-event timing of the whole-MLP kernels (csrc/mlp1d.hip) on the Lorenz local kernel at eval.py's batch (1024 x 61 windows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sda_amd.nn import ResMLP
from sda_amd.utils import ACTIVATIONS
dev = torch.device('cuda:0')
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 61
net = ResMLP(47, 15, hidden_features=[128] * 5, activation=ACTIVATIONS['SiLU']).to(dev)
x = torch.randn(rows, 47, device=dev, requires_grad=True)
g = torch.randn(rows, 15, device=dev)
flops = 2.0 * rows * (47 * 128 + 10 * 128 * 128 + 128 * 15 + 2 * 15 * 15)
for name, fn in (('(clock ramp: discard)', lambda: net(x)), ('fwd (no saves)', lambda: net(x.detach())), ('fwd + saves', lambda: net(x)),):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{name:16s} {ms * 1e3:8.1f} us   {flops / ms / 1e9:7.1f} TFLOP/s ({flops / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak)')
out = net(x)
for _ in range(10): torch.autograd.grad(out, x, g, retain_graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): torch.autograd.grad(out, x, g, retain_graph=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f'{"VJP":16s} {ms * 1e3:8.1f} us   {flops / ms / 1e9:7.1f} TFLOP/s ({flops / ms / 1e9 / 157.3:.3f})')
