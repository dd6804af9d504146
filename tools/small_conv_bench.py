#!/usr/bin/env python3
"""Latency of one small 1-D convolution launch (the Lorenz nets' shape): graph replay of 50 dependent launches,
HIP-event timed.  N / L environment variables set the batch and length; SDA_CONV_DEBUG ablation bits apply."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sda_amd import ops  # noqa: E402
from sda_amd.engine import launch_conv, planar_source  # noqa: E402

dev = torch.device('cuda:0')
n, c, L = int(os.environ.get('N', 1)), 64, int(os.environ.get('L', 64))
x = torch.randn(n, c, 1, L, device=dev)
pk = ops.PackedConv(torch.randn(c, c, 3, device=dev) * 0.05, torch.randn(c, device=dev))
out = torch.empty(n, c, 1, L, device=dev)
res = torch.randn_like(out)
mean, rstd = torch.zeros(n * L, device=dev), torch.ones(n * L, device=dev)
mod = torch.randn(n, c, device=dev)


def run():
    launch_conv(pk, planar_source(x), out, 1, L, circular=False, bias=pk.bias, ln=(mean, rstd), mod=mod, mod_sn=c, res=res)


for _ in range(20):
    run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50):
        run()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f'n={n} L={L} SDA_CONV_DEBUG={os.environ.get("SDA_CONV_DEBUG", "0")}: {e0.elapsed_time(e1):.2f} us per launch')
