#!/usr/bin/env python3
"""Per-layer micro-benchmark of conv_igemm on the reference Kolmogorov net's layer shapes (HIP-event timing).

    python tools/conv_bench.py [--n 896] [--size 64] [--reps 5] [--widths 96,192,384]

--widths 64,128,256: the reference's DEFAULT Kolmogorov widths (experiments/kolmogorov/utils.py:52) instead of the training config's.
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
ap.add_argument('--widths', default='96,192,384')
ap.add_argument('--cin', type=int, default=11, help='input channels of the head (window x state + forcing)')
args = ap.parse_args()
dev = torch.device('cuda:0')
S = args.size
# (name, cin, cout, h_in, stride, up, transpose/zins, fused-LN)
C0, C1, C2 = (int(v) for v in args.widths.split(','))
CI = args.cin
LAYERS = [
    (f'head0   {CI}->{C0}  s1', CI, C0, S, 1, 1, 1, False),
    (f'blk0    {C0}->{C0}  LN', C0, C0, S, 1, 1, 1, True),
    (f'blk0    {C0}->{C0}  act', C0, C0, S, 1, 1, 1, 'act'),
    (f'blk0    {C0}->{C0}  actonly', C0, C0, S, 1, 1, 1, 'actonly'),
    (f'blk0    {C0}->{C0}  resonly', C0, C0, S, 1, 1, 1, 'resonly'),
    (f'blk0    {C0}->{C0}  plain', C0, C0, S, 1, 1, 1, False),
    (f'blk0^T  {C0}->{C0}  dact', C0, C0, S, 1, 1, 1, 'dact'),
    (f'head1   {C0}->{C1} s2', C0, C1, S, 2, 1, 1, False),
    (f'blk1   {C1}->{C1} LN', C1, C1, S // 2, 1, 1, 1, True),
    (f'blk1   {C1}->{C1} act', C1, C1, S // 2, 1, 1, 1, 'act'),
    (f'blk1   {C1}->{C1} plain', C1, C1, S // 2, 1, 1, 1, False),
    (f'blk1^T {C1}->{C1} dact', C1, C1, S // 2, 1, 1, 1, 'dact'),
    (f'head2  {C1}->{C2} s2', C1, C2, S // 2, 2, 1, 1, False),
    (f'blk2   {C2}->{C2} LN', C2, C2, S // 4, 1, 1, 1, True),
    (f'blk2   {C2}->{C2} act', C2, C2, S // 4, 1, 1, 1, 'act'),
    (f'blk2   {C2}->{C2} plain', C2, C2, S // 4, 1, 1, 1, False),
    (f'blk2^T {C2}->{C2} dact', C2, C2, S // 4, 1, 1, 1, 'dact'),
    (f'tail2  {C2}->{C1} up', C2, C1, S // 4, 1, 2, 1, 'tail'),
    (f'tail1  {C1}->{C0}  up', C1, C0, S // 2, 1, 2, 1, 'tail'),
    (f'tail0   {C0}->{CI - 1}    ', C0, CI - 1, S, 1, 1, 1, False),
    (f'head1^T {C1}->{C0} zi', C1, C0, S // 2, 1, 1, 2, False),
]
torch.manual_seed(0)
print(f'n={args.n} size={S} widths={args.widths}')
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
    elif fused == 'tail':                               # LayerNorm -> Upsample -> conv + skip (sda/nn.py:161-169, 203-204)
        mean = torch.zeros(args.n * h * h, device=dev); rstd = torch.ones_like(mean)
        kw.update(ln=(mean, rstd), res=torch.randn_like(out))
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
    print(f'{name:24s} mt={pk.mt} {ms:9.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s  ({flops / 1e9:9.1f} GFLOP)')
print(f'sum {tot_t:.2f} ms  {tot_f / tot_t / 1e9:.1f} TFLOP/s')
