#!/usr/bin/env python3
"""Phase cycles of mlp_fwd_kernel's workgroup 0 (tooling build):
    SDA_LIBDIR=sda_amd/lib_ml SDA_EXTRA_HIPCC_FLAGS=-DSDA_ML_TRACE python -m sda_amd.build
    SDA_HIP_LIB=sda_amd/lib_ml/libsda_hip.so python tools/mlp_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sda_amd import _lib
from sda_amd.nn import ResMLP
from sda_amd.utils import ACTIVATIONS
dev = torch.device('cuda:0')
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 61
net = ResMLP(47, 15, hidden_features=[128] * 5, activation=ACTIVATIONS['SiLU']).to(dev)
x = torch.randn(rows, 47, device=dev, requires_grad=True)
lib = _lib.load()
lib.sda_ml_trace_read.restype = ctypes.c_int
buf = (ctypes.c_longlong * 16)()
for _ in range(5): net(x)
torch.cuda.synchronize()
lib.sda_ml_trace_read(buf, 1)
N = 20
for _ in range(N): net(x)
torch.cuda.synchronize()
lib.sda_ml_trace_read(buf, 0)
names = ['first slab + input rows', 'GEMM set-up (stage descriptor, bias fragments)', 'GEMM (MFMAs, A reads, slab staging)', 'LayerNorm', 'epilogue (saves / act / residual)', 'slab hand-off barrier', 'output rows']
tot = sum(buf[i] for i in range(7)) / N
for i, n in enumerate(names):
    print(f'{n:46s} {buf[i] / N:10.0f} cycles  {100 * buf[i] / N / tot:5.1f} %')
print(f'{"sum":46s} {tot:10.0f} cycles (MFMA floor of a 64-row tile: ~87 000)')
