#!/usr/bin/env python3
"""VERDICT r3 item 5 -- LayerNorm-backward fusion into the epilogue of the first block convolution's VJP at the 96-channel level:
the measurement that decides it.  A fused kernel needs, per output element, TWO epilogue operands (the block input a for x_hat, the
residual cotangent g), ~11 VALU instructions (x_hat, two channel sums, the update) and two reductions across the workgroup's two
consumer halves.  What the existing kernels measure of that cost on the same launch (conv1^T: plain 96 -> 96, backward-data packing):

    a  plain launch                                   (EPM 0: inverse transform + stores)
    b  + residual through the helpers                 (EPM 1: one operand, 1 packed add per pair)
    c  + x act'(z) through the helpers                (EPM 1: one operand, ~30 VALU cycles per value)
    d  + both operands, consumer-side loads           (EPM 2: the two-operand traffic, the only existing two-operand path)
    e  sda_ln_bwd on the same tensors                 (what fusion would remove)

Optimistic model of the fused launch: a + (b - a) + (c - a) -- two helper-fed operands at the cost of one each, LN-backward arithmetic
at the price of SiLU' (it is not cheaper: 11 dependent-free VALU per value against 2 transcendentals + 5 packed ops per PAIR), no charge
for the channel reductions, the second pass over the outputs or the second 48 KiB LDS exchange the stage pipeline has no room for.
Gate (VERDICT): fused <= 0.75 x (a + e).      python tools/lnbwd_fusion_probe.py [--size 256 --n 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sda_amd import ops  # noqa: E402
from sda_amd.engine import launch_conv, planar_source  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--n', type=int, default=30)
args = ap.parse_args()
dev = torch.device('cuda:0')
n, c, h = args.n, 96, args.size
torch.manual_seed(0)
x = torch.randn(n, c, h, h, device=dev)
w = torch.randn(c, c, 3, 3, device=dev) * 0.05
pk = ops.PackedConv(w, None, transpose=True)
out = torch.empty(n, c, h, h, device=dev)
res, z = torch.randn_like(out), torch.randn_like(out)
mean, rstd = torch.randn(n * h * h, device=dev) * 0.1, torch.rand(n * h * h, device=dev) + 0.5
mod = torch.randn(1, c, device=dev)


def timed(fn, warm=25, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


src = planar_source(x)
t = {}
t['a plain'] = timed(lambda: launch_conv(pk, src, out, h, h, circular=True))
t['b + residual (helpers)'] = timed(lambda: launch_conv(pk, src, out, h, h, circular=True, res=res))
t["c + x act'(z) (helpers)"] = timed(lambda: launch_conv(pk, src, out, h, h, circular=True, dact_z=z, act_d=1))
t['d + both (consumer-side loads)'] = timed(lambda: launch_conv(pk, src, out, h, h, circular=True, dact_z=z, act_d=1, res=res))
gx = torch.empty_like(out)
t['e sda_ln_bwd'] = timed(lambda: ops.ln_bwd(out, x, h, h, mod, 0, mean, rstd, True, (1, 1), res, gx))
a, b, cc, d, e = (t[k] for k in t)
fused = b + cc - a
gate = 0.75 * (a + e)
print(f'conv1^T 96 -> 96 at {h}^2, {n} windows (ms per launch, HIP events, 25 warm-up launches):')
for k, v in t.items():
    print(f'  {k:34s} {v:8.3f}')
print(f'  two-kernel form a + e              {a + e:8.3f}')
print(f'  optimistic fused model b + c - a   {fused:8.3f}   ({100 * (fused / (a + e) - 1):+.1f} % vs a + e)')
print(f'  existing two-operand launch d      {d:8.3f}   ({100 * (d / (a + e) - 1):+.1f} % vs a + e)')
print(f'  gate 0.75 (a + e)                  {gate:8.3f}   -> {"GO" if fused <= gate else "NO-GO"}')
bytes_ln = 4.0 * n * h * h * (c * 4 + 2)
print(f'  sda_ln_bwd: {bytes_ln / e / 1e6:.0f} GB/s of its algorithmic bytes')
