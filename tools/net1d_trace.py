#!/usr/bin/env python3
"""Phase cycles of net1d_fwd_kernel (workgroup 0 / wave 0, averaged over launches) next to the HIP-event time per launch, for the
Lorenz-96 / Lorenz-63 shapes.  Builds its own tooling copy of csrc/net1d.hip with -DSDA_N1_TRACE (the product library carries no
stamps).      python tools/net1d_trace.py"""
import ctypes, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import torch.nn as nn
from sda_amd import _lib, ops
from sda_amd.nn import UNet
so = '/tmp/libn1trace.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-DSDA_N1_TRACE'] + os.environ.get('N1_FLAGS', '').split() + [
                       '-shared', os.path.join(R, 'sda_amd/csrc/net1d.hip'), '-o', so])
_lib.load()
tl = ctypes.CDLL(so)
tl.sda_net1d_fwd.argtypes = [ctypes.POINTER(_lib.Net1dDesc), ctypes.c_void_p]
tl.sda_n1_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device('cuda:0')
names = ['issue', 'x round trip', 'head mm', 'blk: operands + a_save + reduce 1', 'blk: reduce 2 + tile', 'blk: conv1 mm', 'blk: conv1 epilogue',
         'blk: conv2 mm', 'blk: residual', 'tail + stores']
for n, cin, L in ((64, 40, 128), (1, 3, 64)):
    net = UNet(cin, cin, 32, hidden_channels=(64,), hidden_blocks=(3,), kernel_size=3, activation=nn.SiLU, spatial=1, padding_mode='zeros').to(dev)
    eng = net.engine()
    x = torch.randn(n, L, cin, device=dev).transpose(1, 2)
    from sda_amd.engine import source_from_tensor
    xv, src = source_from_tensor(x, 1)
    plan = eng.net1d_plan(src)
    mod_all = eng.modulation(torch.randn(1, 32, device=dev))
    out = torch.empty(n, 1, L, cin, device=dev).permute(0, 3, 1, 2)
    for save in (False, True):
        d, keep = eng._net1d_desc(plan, n, L, mod_all, 0, False, False)
        d.x = src.x.data_ptr(); d.x_sn, d.x_sc, d.x_sx = src.sn_outer, src.sc, src.sx
        d.out = out.data_ptr(); d.out_sn, d.out_sc, d.out_sx = out.stride(0), out.stride(1), out.stride(3)
        if save:
            a_s = torch.empty(6, n, 64, L, device=dev); z_s = torch.empty_like(a_s)
            m_s = torch.empty(6, n, L, device=dev); r_s = torch.empty_like(m_s)
            d.a_save, d.z_save, d.save_stride = a_s.data_ptr(), z_s.data_ptr(), n * 64 * L
            d.mean_save, d.rstd_save, d.stat_stride = m_s.data_ptr(), r_s.data_ptr(), n * L
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: tl.sda_net1d_fwd(ctypes.byref(d), st)
        for _ in range(20): assert run() == 0
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)()
        tl.sda_n1_trace_read(ctypes.cast(buf, ctypes.c_void_p), 1)
        N = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N): run()
        e1.record(); torch.cuda.synchronize()
        tl.sda_n1_trace_read(ctypes.cast(buf, ctypes.c_void_p), 1)
        tot = sum(buf[k] for k in range(10)) / N
        print(f'n={n} cin={cin} L={L} save={save}: {e0.elapsed_time(e1) / N * 1e3:.1f} us per launch (eager, back to back); {tot:.0f} cycles in workgroup 0:')
        for k in range(10):
            per = buf[k] / N / (6 if 3 <= k <= 8 else 1)
            print(f'    {names[k]:36s} {buf[k] / N:9.0f}' + (f'   ({per:7.0f} per block)' if 3 <= k <= 8 else ''))
