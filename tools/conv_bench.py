#!/usr/bin/env python3
"""Per-layer micro-benchmark of conv_igemm on the reference Kolmogorov net's layer shapes (HIP-event timing).

    python tools/conv_bench.py [--n 896] [--size 64] [--reps 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sda_amd import ops  # noqa: E402
from sda_amd.engine import launch_conv, planar_source  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=896)
ap.add_argument('--size', type=int, default=64)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--only', default='')
args = ap.parse_args()
dev = torch.device('cuda:0')
S = args.size
# (name, cin, cout, h_in, stride, up, transpose/zins, fused-LN)
LAYERS = [
    ('head0   11->96  s1', 11, 96, S, 1, 1, 1, False),
    ('blk0    96->96  LN', 96, 96, S, 1, 1, 1, True),
    ('blk0    96->96  act', 96, 96, S, 1, 1, 1, 'act'),
    ('blk0    96->96  actonly', 96, 96, S, 1, 1, 1, 'actonly'),
    ('blk0    96->96  resonly', 96, 96, S, 1, 1, 1, 'resonly'),
    ('blk0    96->96  plain', 96, 96, S, 1, 1, 1, False),
    ('blk0^T  96->96  dact', 96, 96, S, 1, 1, 1, 'dact'),
    ('head1   96->192 s2', 96, 192, S, 2, 1, 1, False),
    ('blk1   192->192 LN', 192, 192, S // 2, 1, 1, 1, True),
    ('head2  192->384 s2', 192, 384, S // 2, 2, 1, 1, False),
    ('blk2   384->384 LN', 384, 384, S // 4, 1, 1, 1, True),
    ('tail2  384->192 up', 384, 192, S // 4, 1, 2, 1, True),
    ('tail1  192->96  up', 192, 96, S // 2, 1, 2, 1, True),
    ('tail0   96->10    ', 96, 10, S, 1, 1, 1, False),
    ('head1^T 192->96 zi', 192, 96, S // 2, 1, 1, 2, False),
]
torch.manual_seed(0)
print(f'n={args.n} size={S}')
tot_f, tot_t = 0.0, 0.0
for name, cin, cout, h, stride, up, zins, fused in LAYERS:
    if args.only and args.only not in name:
        continue
    x = torch.randn(args.n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    pk = ops.PackedConv(w, b)
    ho = h * up * zins // stride
    out = torch.empty(args.n, cout, ho, ho, device=dev)
    kw = dict(circular=True, stride=(stride, stride), up=(up, up), zins=(zins, zins), bias=pk.bias)
    if fused is True:
        mean = torch.zeros(args.n * h * h, device=dev); rstd = torch.ones_like(mean)
        kw.update(ln=(mean, rstd), mod=torch.randn(1, cin, device=dev))
    elif fused == 'act':
        kw.update(act_in=1, res=torch.randn_like(out))
    elif fused == 'actonly':
        kw.update(act_in=1)
    elif fused == 'resonly':
        kw.update(res=torch.randn_like(out))
    elif fused == 'dact':
        kw.update(dact_z=torch.randn_like(out), act_d=1)
    launch_conv(pk, planar_source(x), out, ho, ho, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        launch_conv(pk, planar_source(x), out, ho, ho, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    flops = 2.0 * args.n * (ho * ho / zins ** 2) * cout * cin * 9
    tot_f += flops; tot_t += ms
    print(f'{name:22s} mt={pk.mt} {ms:9.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s  ({flops / 1e9:9.1f} GFLOP)')
print(f'sum {tot_t:.2f} ms  {tot_f / tot_t / 1e9:.1f} TFLOP/s')
