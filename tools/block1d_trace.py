#!/usr/bin/env python3
"""Phase cycles of block1d_fwd_kernel (build with SDA_EXTRA_HIPCC_FLAGS=-DSDA_B1_TRACE): workgroup 0 / wave 0, averaged over launches,
next to the HIP-event time per launch.    python tools/block1d_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sda_amd import _lib, ops
dev = torch.device('cuda:0')
lib = _lib.load()
for n, c, L in ((64, 64, 128), (1, 64, 64)):
    a = torch.randn(n, c, 1, L, device=dev)
    mod = torch.randn(1, c, device=dev)
    w1 = torch.randn(c, c, 3, device=dev) * 0.05; w2 = torch.randn(c, c, 3, device=dev) * 0.05
    pk1 = ops.PackedConv(w1, torch.randn(c, device=dev)); pk2 = ops.PackedConv(w2, torch.randn(c, device=dev))
    y = torch.empty_like(a); z = torch.empty_like(a)
    mean = torch.empty(n * L, device=dev); rstd = torch.empty_like(mean)
    def run():
        ops.block1d_fwd(a, mod, 0, pk1, pk2, False, 1, 1e-5, True, y, z, mean, rstd)
    for _ in range(20): run()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 16)()
    lib.sda_b1_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sda_b1_trace_read(ctypes.cast(buf, ctypes.c_void_p), 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 200
    e0.record()
    for _ in range(N): run()
    e1.record(); torch.cuda.synchronize()
    lib.sda_b1_trace_read(ctypes.cast(buf, ctypes.c_void_p), 1)
    print(f'n={n} c={c} L={L}: {e0.elapsed_time(e1) / N * 1e3:.1f} us per launch (eager, back to back); cycles per phase '
          f'[issue loads, round trip + 1st reduction, 2nd reduction + tile, conv1 + z, conv2 + y]:', [buf[k] // N for k in range(5)])
