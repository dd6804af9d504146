#!/usr/bin/env python3
"""VERDICT r3 item 7, second half: what conv_wino4's PIPELINE (helpers: halo loads, loader fusions, B^T d B, U / V through LDS; two
workgroup barriers per stage; epilogue) delivers when the multiply gets 2x / 3x cheaper -- the bracket around the 2.67x of a six-product
bf16 multiply -- with and without the ~8 VALU instructions per V value that splitting costs the helpers.  Tooling build only (results of
the variants are wrong: they skip MFMAs):

    SDA_LIBDIR=sda_amd/lib_w4v SDA_EXTRA_HIPCC_FLAGS=-DSDA_W4_VARIANTS python -m sda_amd.build
    SDA_HIP_LIB=sda_amd/lib_w4v/libsda_hip.so python tools/bf16x6_pipeline_probe.py
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == '--one':
    import torch
    from sda_amd import ops
    from sda_amd.engine import launch_conv, planar_source
    dev = torch.device('cuda:0')
    out = []
    for name, c, h, n in (('96->96 @256^2', 96, 256, 30), ('384->384 @64^2', 384, 64, 120), ('96->96 @64^2', 96, 64, 896)):
        x = torch.randn(n, c, h, h, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        pk = ops.PackedConv(w, None)
        o = torch.empty(n, c, h, h, device=dev)
        src = planar_source(x)
        for _ in range(25):
            launch_conv(pk, src, o, h, h, circular=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            launch_conv(pk, src, o, h, h, circular=True)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10)
    print(' '.join(f'{v:.4f}' for v in out))
    sys.exit(0)

rows = {}
for var, label in ((0, 'shipped kernel (96 fp32 MFMAs per stage)'), (12, 'multiply 2x cheaper (48 MFMAs)'), (13, 'multiply 3x cheaper (32 MFMAs)'),
                   (14, 'multiply 2x cheaper + split VALU in the helpers'), (15, 'multiply 3x cheaper + split VALU in the helpers')):
    env = dict(os.environ, SDA_W4_VAR=str(var))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--one'], env=env, capture_output=True, text=True)
    vals = [float(v) for v in r.stdout.strip().splitlines()[-1].split()] if r.returncode == 0 and r.stdout.strip() else None
    rows[var] = (label, vals)
    if vals is None:
        print(label, 'FAILED', r.stderr[-400:])
base = rows[0][1]
print('plain (no loader fusion, no epilogue operand) conv_wino4 launches, ms per launch and speed-up over the shipped kernel:')
print(f'{"":52s} {"96->96 @256^2 x30":>22s} {"384->384 @64^2 x120":>22s} {"96->96 @64^2 x896":>22s}')
for var, (label, vals) in rows.items():
    if vals:
        print(f'{label:52s} ' + ' '.join(f'{v:9.3f} ms ({b / v:4.2f}x)' for v, b in zip(vals, base)))
