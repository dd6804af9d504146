#!/usr/bin/env python3
"""Phase cycles of step1d_prologue_kernel (tooling build):
    SDA_LIBDIR=sda_amd/lib_s1 SDA_EXTRA_HIPCC_FLAGS=-DSDA_S1_TRACE python -m sda_amd.build
    SDA_HIP_LIB=sda_amd/lib_s1/libsda_hip.so python tools/step1d_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sda_amd import fused1d, observe as Ob
from sda_amd.experiments.lorenz import make_global_score
from sda_amd.score import GaussianScore, VPSDE
dev = torch.device('cuda:0')
net = make_global_score().to(dev)
inner = VPSDE(net, shape=())
gs = GaussianScore(torch.randn(8, 1), A=Ob.Subsample((slice(None, None, 8), slice(0, 1))), std=0.1, sde=inner).to(dev)
x = torch.randn(1, 64, 3, device=dev)
fz = fused1d.plan(gs, x, torch.tensor(0.5, device=dev), None)
table = torch.rand(300, 5, device=dev)
istep = torch.zeros(1, device=dev, dtype=torch.int64)
acc = torch.zeros(7)
for i in range(60):
    fz.prologue_step(table, istep)
    torch.cuda.synchronize()
    if i >= 10:
        acc += fz.coef[9:16].cpu()
names = ['staging issued + scalars', 'staging landed (barrier)', 'features', 'hidden layer', 'embedding', 'projection', '-']
prev = 0.0
for n, v in zip(names, (acc / 50).tolist()):
    print(f'{n:28s} {v:9.0f} cycles (+{v - prev:7.0f})')
    prev = v
