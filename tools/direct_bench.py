#!/usr/bin/env python3
"""HIP-event timings of the NON-Winograd convolutions of the Kolmogorov net at the configs[3] scale (the `direct` family of
bench.py's roofline: stride-2 level heads, their parity-split VJPs, the 10-channel tail / head^T), issued flops over the fp32
matrix peak.        python tools/direct_bench.py [n_windows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from sda_amd import ops
from sda_amd.engine import _ConvCache, launch_conv, planar_source
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
S = 256


def timeit(fn, warm=10, reps=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, flops):
    print(f'{name:44s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF  (mfma util {flops / ms / 1e9 / 157.3:.2f})', flush=True)


total_ms, total_fl = 0.0, 0.0
for lvl, (cin, cout, s_in) in enumerate(((96, 192, S), (192, 384, S // 2)), start=1):
    conv = nn.Conv2d(cin, cout, 3, stride=2, padding=1, padding_mode='circular').to(dev)
    cc = _ConvCache(conv)
    x = torch.randn(n, cin, s_in, s_in, device=dev)
    so = s_in // 2
    y = torch.empty(n, cout, so, so, device=dev)
    pk = cc.fwd()
    fl = 2.0 * n * so * so * cout * cin * 9
    ms = timeit(lambda: launch_conv(pk, planar_source(x), y, so, so, circular=True, stride=(2, 2), bias=pk.bias))
    report(f'head{lvl} fwd {cin}->{cout} s2 @{s_in}', ms, fl); total_ms += ms; total_fl += fl
    g = torch.randn(n, cout, so, so, device=dev)
    g2 = torch.empty(n, cin, s_in, s_in, device=dev)
    skip = torch.randn_like(g2)
    tot = 0.0
    for py, px, pkp, pad in cc.bwd_parity():
        view = g2[:, :, py::2, px::2]
        ms = timeit(lambda: launch_conv(pkp, planar_source(g), view, view.shape[2], view.shape[3], circular=True, pad=pad, res=skip[:, :, py::2, px::2]))
        taps = pkp.kh * pkp.kw
        report(f'head{lvl}^T parity ({py},{px}) {pkp.kh}x{pkp.kw} {cout}->{cin}', ms, 2.0 * n * so * so * cout * cin * taps)
        tot += ms
    report(f'head{lvl}^T all four classes (4 launches)', tot, fl)
    w4 = cc.bwd_parity4()
    cls = cc.bwd_parity()
    v00, s00 = g2[:, :, 0::2, 0::2], skip[:, :, 0::2, 0::2]
    if w4 is not None and launch_conv(cls[0][2], planar_source(g), v00, v00.shape[2], v00.shape[3], circular=True, pad=cls[0][3], res=s00, parity4_w=w4) is not None:
        tot = timeit(lambda: launch_conv(cls[0][2], planar_source(g), v00, v00.shape[2], v00.shape[3], circular=True, pad=cls[0][3], res=s00, parity4_w=w4))
        report(f'head{lvl}^T one launch (conv_par4)', tot, fl)
    total_ms += tot; total_fl += fl
# tail0: 96 -> 10 and head0^T: 96 -> 10 (VJP of the 11 -> 96 head, forcing-channel gradient dropped)
conv = nn.Conv2d(96, 10, 3, padding=1, padding_mode='circular').to(dev)
cc = _ConvCache(conv)
x = torch.randn(n, 96, S, S, device=dev)
y = torch.empty(n, 10, S, S, device=dev)
pk = cc.fwd()
fl = 2.0 * n * S * S * 10 * 96 * 9
ms = timeit(lambda: launch_conv(pk, planar_source(x), y, S, S, circular=True, bias=pk.bias))
report('tail0 96->10 @256', ms, fl); total_ms += ms; total_fl += fl
conv = nn.Conv2d(11, 96, 3, padding=1, padding_mode='circular').to(dev)
cc = _ConvCache(conv)
pkb = cc.bwd(cin_keep=10)
ms = timeit(lambda: launch_conv(pkb, planar_source(x), y, S, S, circular=True))
report('head0^T 96->10 @256', ms, fl); total_ms += ms; total_fl += fl
report('direct family, one net evaluation fwd + VJP', total_ms, total_fl)
