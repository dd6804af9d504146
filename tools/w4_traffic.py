#!/usr/bin/env python3
"""Workload for the per-layer HBM-traffic measurement of conv_wino4 (run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`
or `WRITE_SIZE`; tools/traffic.sh drives it, tools/w4_traffic_post.py reads the counter CSVs).

Launches each K64 layer shape at the 256^2 scale K times in a fixed order and prints that order with the ALGORITHMIC bytes of
a launch (every operand read once, the output written once) -- the post-processor assigns the conv_wino4 dispatches of the
trace to the layers by position.  The batch (windows per launch) is sized so that a layer's input alone exceeds the 256 MiB
Infinity Cache several times over."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sda_amd import ops  # noqa: E402
from sda_amd.engine import launch_conv, planar_source  # noqa: E402

dev = torch.device('cuda:0')
K = int(os.environ.get('W4T_K', '3'))
S = int(os.environ.get('W4T_S', '256'))
N = int(os.environ.get('W4T_N', '30'))
LAYERS = [('blk0 96->96 plain', 96, 96, S, {}), ('blk0 96->96 mod+LN', 96, 96, S, dict(ln=True, mod=True)),
          ('blk0 96->96 silu+res', 96, 96, S, dict(silu=True, res=True)), ('blk0^T 96->96 dact', 96, 96, S, dict(dact=True)),
          ('blk1 192->192 mod+LN', 192, 192, S // 2, dict(ln=True, mod=True)), ('blk1 192->192 silu+res', 192, 192, S // 2, dict(silu=True, res=True)),
          ('blk2 384->384 mod+LN', 384, 384, S // 4, dict(ln=True, mod=True)), ('blk2 384->384 silu+res', 384, 384, S // 4, dict(silu=True, res=True)),
          ('tail1 192->96 up+LN+res', 192, 96, S, dict(ln=True, up=True, res=True)), ('tail2 384->192 up+LN+res', 384, 192, S // 2, dict(ln=True, up=True, res=True))]
order = []
for name, cin, cout, h, fz in LAYERS:
    n = N
    hs = h // 2 if fz.get('up') else h
    x = torch.randn(n, cin, hs, hs, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    pk = ops.PackedConv(w, b)
    out = torch.empty(n, cout, h, h, device=dev)
    kw = dict(circular=True, bias=pk.bias)
    rd = 4.0 * n * cin * hs * hs + 4.0 * 16 * cin * cout            # input once + the transformed weights once
    if fz.get('up'):
        kw['up'] = (2, 2)
    if fz.get('ln'):
        kw['ln'] = (torch.zeros(n * hs * hs, device=dev), torch.ones(n * hs * hs, device=dev))
        rd += 8.0 * n * hs * hs
    if fz.get('mod'):
        kw['mod'] = torch.randn(1, cin, device=dev)
    if fz.get('silu'):
        kw['act_in'] = 1
    if fz.get('res'):
        kw['res'] = torch.randn_like(out)
        rd += 4.0 * out.numel()
    if fz.get('dact'):
        kw.update(dact_z=torch.randn_like(out), act_d=1)
        rd += 4.0 * out.numel()
    if ops.MULTIPLY == 'f16x2' and not fz.get('ln'):                  # (conv_h2 needs the input's scale: randn * 1 stays below 6.5)
        kw['x_amax'] = torch.full((1,), 6.5, device=dev)
    d = launch_conv(pk, planar_source(x), out, h, h, **kw)
    assert d.w_h2 or ops.conv_path(d) in (2, 5), name
    for _ in range(K - 1):
        launch_conv(pk, planar_source(x), out, h, h, **kw)
    torch.cuda.synchronize()
    order.append(dict(layer=name, launches=K, alg_read_bytes=rd, alg_write_bytes=4.0 * out.numel(),
                      flops=2.0 * n * h * h * cout * cin * 9))
    del x, out, kw
    torch.cuda.empty_cache()
print('W4T_ORDER ' + json.dumps(order), flush=True)
