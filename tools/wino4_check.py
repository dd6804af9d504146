#!/usr/bin/env python3
"""Correctness + speed of the one-wave-per-SIMD Winograd kernel (conv_wino4.hip) on the GPU box.

    python tools/wino4_check.py [--cases 120] [--bench]

1. structured cases (every loader / epilogue fusion, both paddings, upsampling, partial K-stages, several cout tiles,
   batches that leave workgroups with 0 / 1 / many tiles) + a randomised sweep, each against torch float64;
2. --bench: the K64 layer shapes at 64^2 and 256^2 with both Winograd kernels (SDA_CONV_WINO4 toggled per process is not
   possible, so the old kernel is timed by passing descriptors without w_wino4)."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sda_amd import ops  # noqa: E402
from sda_amd.engine import launch_conv, planar_source  # noqa: E402

dev = torch.device('cuda:0')


def ref_conv(x, w, b, circular):
    xp = F.pad(x, (1, 1, 1, 1), mode='circular' if circular else 'constant')
    return F.conv2d(xp, w, b)


def run_case(n, cin, cout, h, w_, circular, mod, ln, silu, up, dact, res, bias, seed, transpose=False):
    g = torch.Generator().manual_seed(seed)
    hs, ws = (h // 2, w_ // 2) if up else (h, w_)
    x = torch.randn(n, cin, hs, ws, generator=g) * 1.3 + 0.1
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    xin, opts = x.double(), {}
    if mod:
        m = torch.randn(1, cin, generator=g)
        opts['mod'] = m
        xin = xin + m.double()[:, :, None, None]
    if ln:
        var, mean = torch.var_mean(xin, dim=1, unbiased=True, keepdim=True)
        rstd = 1 / torch.sqrt(var + 1e-5)
        opts['ln'] = (mean.float().reshape(n, -1), rstd.float().reshape(n, -1))
        xin = (xin - mean) * rstd
    if silu:
        opts['act_in'] = 1
        xin = F.silu(xin)
    if up:
        opts['up'] = (2, 2)
        xin = xin.repeat_interleave(2, -1).repeat_interleave(2, -2)
    ref = ref_conv(xin, wgt.double(), None if b is None else b.double(), circular)
    if dact:
        z = torch.randn(ref.shape, generator=g)
        zz = z.double().requires_grad_(True)
        dz, = torch.autograd.grad(F.silu(zz).sum(), zz)
        ref = ref * dz
        opts['dact_z'], opts['act_d'] = z, 1
    if res:
        r = torch.randn(ref.shape, generator=g)
        ref = ref + r.double()
        opts['res'] = r
    pk = ops.PackedConv(wgt.to(dev), None if b is None else b.to(dev))
    out = torch.full((n, cout, h, w_), float('nan'), device=dev)
    dopts = {}
    for k, v in opts.items():
        dopts[k] = tuple(t.to(dev).contiguous() for t in v) if k == 'ln' else (v.to(dev).contiguous() if torch.is_tensor(v) else v)
    xd = x.to(dev)
    desc = launch_conv(pk, planar_source(xd), out, h, w_, circular=circular, bias=pk.bias, **dopts)
    torch.cuda.synchronize()
    path = ops.conv_path(desc)
    got = out.cpu().double()
    nan = int(torch.isnan(got).sum())
    scale = ref.abs().max().item() + 1e-30
    diff = (got - ref).abs()
    diff[torch.isnan(diff)] = float('inf')
    err = diff.max().item() / scale
    info = ''
    if err > 1e-4:
        bad = (diff > 1e-4 * scale)
        idx = bad.nonzero()
        info = (f' bad={int(bad.sum())}/{bad.numel()} nan={nan} first={idx[0].tolist()} last={idx[-1].tolist()} '
                f'bad per image={bad.flatten(1).sum(1).tolist()[:6]} per cout(first 12)={bad.sum((0, 2, 3)).tolist()[:12]} '
                f'rows={bad.sum((0, 1, 3)).tolist()[:20]} cols={bad.sum((0, 1, 2)).tolist()[:20]}')
    return path, err, info


def expect_path(c):
    """the second-generation kernel serves the four loader configurations of the reference U-Net; the first generation the rest --
    where it exists: cout % 96 == 0.  Widths that are multiples of 64 only run the 64-cout tile of conv_wino4 (MF = 2) and fall
    back to the direct kernel (0)."""
    key = (bool(c['mod']), bool(c['ln']), bool(c['silu']))
    mf = 3 if c['cout'] % 96 == 0 else (2 if c['cout'] % 64 == 0 else 1)
    nstage = (c['cin'] + 7) // 8
    # the up-sampled zero-position form: at 96 couts where the operand goes through the helpers (twelve stages on), at 64 couts always
    # (consumer-side loads), never at 32
    epi_ok = (mf == 3 and nstage >= 12) or mf == 2
    if key not in ((False, False, False), (False, False, True), (False, True, False), (True, True, False)):
        return 1 if mf == 3 else 0
    # (5 = its zero-position form: 2 x 2 up-sampled source with the LayerNorm loader and ONE epilogue operand through the helpers -- the skip
    #  tensor in the reference tails; an act'(z) operand alone selects it as well)
    if key == (False, True, False) and c['up'] and (bool(c['res']) != bool(c['dact'])) and epi_ok and os.environ.get('SDA_W4_ZP', '1') != '0':
        return 5
    return 2


def structured():
    cases = []
    base = dict(n=2, cin=16, cout=96, h=8, w_=16, circular=True, mod=False, ln=False, silu=False, up=False, dact=False,
                res=False, bias=False)
    def add(**kw):
        c = dict(base); c.update(kw); cases.append(c)
    add()
    add(circular=False)
    add(bias=True)
    add(cin=32)
    add(cin=96, h=16, w_=32)
    add(cin=24)                      # partial last stage
    add(cin=3)
    add(cout=192)
    add(cout=384, cin=48)
    add(n=1, h=64, w_=64, cin=96)
    add(n=7, h=16, w_=16, cin=40, circular=False)
    add(mod=True)
    add(ln=True)
    add(mod=True, ln=True, cin=96, h=16, w_=32)
    add(silu=True, res=True, cin=96, h=16, w_=32)
    add(dact=True, bias=True)
    add(res=True, dact=True, bias=True, cin=32, circular=False)
    add(up=True, ln=True, cin=32, h=16, w_=32)
    add(up=True, ln=True, cin=192, cout=96, h=32, w_=32, circular=False)
    add(n=300, cin=16, h=8, w_=16)   # more tiles than workgroups, uneven split
    add(n=33, cin=96, cout=192, h=16, w_=16, mod=True, ln=True)
    # ---- the 64-cout tile (MF = 2): the reference's default widths (64, 128, 256) and other multiples of 64
    add(cout=64)
    add(cout=64, circular=False, bias=True)
    add(cout=128, cin=24)            # two cout tiles, partial last stage
    add(cout=256, cin=64, h=16, w_=16)
    add(cout=320, cin=40)            # five cout tiles
    add(cout=64, cin=64, h=64, w_=64, n=3, mod=True, ln=True, bias=True)
    add(cout=64, cin=64, h=16, w_=32, silu=True, res=True, bias=True)          # eight stages: operand through the helpers
    add(cout=64, cin=64, h=16, w_=32, dact=True)
    add(cout=64, cin=56, h=16, w_=32, silu=True, res=True)                      # seven stages: consumer-side loads
    add(cout=64, cin=56, h=16, w_=32, dact=True, circular=False)
    add(cout=128, cin=128, h=32, w_=32, n=5, silu=True, res=True, bias=True)
    add(cout=128, cin=128, h=32, w_=32, n=5, dact=True)
    add(cout=256, cin=256, h=16, w_=16, n=9, dact=True, circular=False)
    add(cout=256, cin=256, h=16, w_=16, n=9, silu=True, res=True)
    add(cout=128, cin=100, h=16, w_=16, res=True, dact=True, bias=True)        # two operands: consumer-side loads
    add(cout=64, cin=128, h=32, w_=32, up=True, ln=True, res=True, bias=True)  # the tail 128 -> 64: zero-position form
    add(cout=128, cin=256, h=16, w_=32, up=True, ln=True, res=True, circular=False)
    add(cout=64, cin=40, h=16, w_=32, up=True, ln=True, res=True)              # five stages: the zero-position form all the same (consumer-side loads)
    add(cout=64, cin=16, h=8, w_=16, n=300)
    add(cout=128, cin=64, h=16, w_=16, n=67, mod=True, ln=True)
    # ---- the 32-cout tile (MF = 1): UNet's own default widths (32, 64, 128), sda/nn.py:99, and other multiples of 32
    add(cout=32)
    add(cout=32, cin=32, h=64, w_=64, n=3, mod=True, ln=True, bias=True)
    add(cout=32, cin=32, h=16, w_=32, silu=True, res=True, bias=True, circular=False)
    add(cout=32, cin=32, h=16, w_=32, dact=True)
    add(cout=160, cin=24, res=True, dact=True)             # five cout tiles, both operands
    add(cout=32, cin=64, h=32, w_=32, up=True, ln=True, res=True, bias=True)   # the tail 64 -> 32: full kernel, consumer-side skip loads
    add(cout=32, cin=16, h=8, w_=16, n=300)
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=120)
    ap.add_argument('--bench', action='store_true')
    ap.add_argument('--variants', default='', help='comma list of SDA_W4_VAR values to time on the plain layers (needs a -DSDA_W4_VARIANTS build)')
    ap.add_argument('--skip-check', action='store_true')
    args = ap.parse_args()
    if os.environ.get('W4_CLOCK'):
        clock_ramp()
    if os.environ.get('W4_ABLATE'):
        ablate()
    if args.variants:
        variants([int(v) for v in args.variants.split(',') if int(v) != 11])
        if '11' in args.variants.split(','):
            trace()
    if args.skip_check:
        if args.bench:
            bench()
        return
    bad = 0
    for i, c in enumerate(structured()):
        path, err, info = run_case(seed=100 + i, **c)
        ok = err <= 1e-4 and path == expect_path(c)
        bad += not ok
        print(f'{"ok  " if ok else "FAIL"} struct {i:2d} path={path} err={err:.2e} {c if not ok else ""}{info}', flush=True)
    rng = random.Random(0)
    worst = 0.0
    for i in range(args.cases):
        c = dict(n=rng.choice([1, 2, 3, 5, 9]), cin=rng.choice([3, 8, 16, 24, 40, 56, 64, 96, 100, 128, 192, 256, 384]),
                 cout=rng.choice([96, 96, 192, 384, 64, 64, 128, 256, 320, 32, 160]), h=rng.choice([8, 16, 24, 32, 64]), w_=rng.choice([16, 32, 48, 64]),
                 circular=rng.random() < 0.6, mod=rng.random() < 0.4, ln=rng.random() < 0.4, silu=rng.random() < 0.4,
                 up=rng.random() < 0.25, dact=rng.random() < 0.3, res=rng.random() < 0.4, bias=rng.random() < 0.6)
        if c['cin'] * c['h'] * c['w_'] * c['n'] > 4e6:
            c['n'] = 1
        path, err, info = run_case(seed=1000 + i, **c)
        ok = err <= 1e-4 and path == expect_path(c)
        worst = max(worst, err if err == err else 1.0)
        if not ok:
            bad += 1
            print(f'FAIL rand {i} path={path} err={err:.2e} {c}{info}', flush=True)
    print(f'random sweep: {args.cases} cases, worst rel err {worst:.2e}; total failures {bad}', flush=True)
    if args.bench:
        bench()
    sys.exit(1 if bad else 0)


def variants(vs):
    print('--- variants on plain layers (algorithmic TFLOP/s | mfma util)')
    for name, cin, cout, h, n in (('96->96 @64', 96, 96, 64, 896), ('192->192 @32', 192, 192, 32, 896),
                                  ('384->384 @16', 384, 384, 16, 896), ('384->384 @64', 384, 384, 64, 60)):
        x = torch.randn(n, cin, h, h, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        pk = ops.PackedConv(w, None)
        out = torch.empty(n, cout, h, h, device=dev)
        row = []
        for v in vs:
            os.environ['SDA_W4_VAR'] = str(v)
            launch_conv(pk, planar_source(x), out, h, h, circular=True)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    launch_conv(pk, planar_source(x), out, h, h, circular=True)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 4)
            tf = 2.0 * n * h * h * cout * cin * 9 / best / 1e9
            row.append(f'v{v}: {tf:6.1f}|{tf / 2.25 / 157.3:.2f}')
        os.environ['SDA_W4_VAR'] = '0'
        print(f'{name:16s} ' + '  '.join(row), flush=True)


def ablate():
    """time the plain 96->96 @64 layer under the runtime ablation switches of a -DSDA_W4_VARIANTS build (results are
    garbage with most of them: timing only) -> cycles per K-stage at the nominal 2.4 GHz"""
    combos = [int(v) for v in os.environ.get('W4_ABLATE', '0').split(',')]
    print('--- ablation (bits: 16 U store, 32 transform, 64 patch reads, 128 raw commit, 256 global loads, 512 no M barrier, '
          '1024 no consumer LDS reads): ms | cycles per stage @2.4 GHz')
    for name, cin, cout, h, n in (('96->96 @64', 96, 96, 64, 896), ('384->384 @16', 384, 384, 16, 896)):
        x = torch.randn(n, cin, h, h, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        pk = ops.PackedConv(w, None)
        out = torch.empty(n, cout, h, h, device=dev)
        stages = n * (h // 8) * (h // 16) * (cout // 96) * (cin // 8) / 256
        for dbg in combos:
            os.environ['SDA_CONV_DEBUG'] = str(dbg)
            for _ in range(2):
                launch_conv(pk, planar_source(x), out, h, h, circular=True)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    launch_conv(pk, planar_source(x), out, h, h, circular=True)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 4)
            print(f'{name:14s} debug={dbg:5d}: {best:7.3f} ms  {best * 2.4e6 / stages:7.0f} cycles/stage', flush=True)
        os.environ['SDA_CONV_DEBUG'] = '0'


def clock_ramp():
    """shader clock over consecutive launches (total cycles of the tracing variant without stamps / launch wall time)"""
    import ctypes
    from sda_amd import _lib
    lib = _lib.load()
    lib.sda_w4_trace_read.argtypes = [ctypes.c_void_p]
    os.environ['SDA_CONV_DEBUG'] = '4096'
    os.environ['SDA_W4_VAR'] = '11'
    cin = cout = 96; h = 64; n = 896
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    pk = ops.PackedConv(w, None)
    out = torch.empty(n, cout, h, h, device=dev)
    buf = (ctypes.c_double * 64)()
    rows = []
    for i in range(60):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch_conv(pk, planar_source(x), out, h, h, circular=True)
        e1.record()
        torch.cuda.synchronize()
        lib.sda_w4_trace_read(ctypes.cast(buf, ctypes.c_void_p))
        rows.append((e0.elapsed_time(e1), buf[7]))
    print('--- clock ramp, 96->96 @64 x 60 launches: ms | cycles per workgroup | GHz')
    for i, (ms, cyc) in enumerate(rows):
        if i < 10 or i % 10 == 9:
            print(f'   launch {i:2d}: {ms:.3f} ms  {cyc:.0f} cycles  {cyc / ms / 1e6:.2f} GHz')
    os.environ['SDA_CONV_DEBUG'] = '0'
    os.environ['SDA_W4_VAR'] = '0'


def trace():
    """phase cycle sums per wave role (SDA_W4_VAR=11 of a -DSDA_W4_VARIANTS build), plain 96->96 @64"""
    import ctypes
    from sda_amd import _lib
    lib = _lib.load()
    dbgs = [int(v) for v in os.environ.get('W4_TRACE_DEBUG', '0').split(',')]
    for dbg in dbgs:
      os.environ['SDA_CONV_DEBUG'] = str(dbg)
      print(f'=== helper skips (debug bits: 16 U store, 32 transform, 64 patch reads, 128 raw commit) = {dbg}')
      cases = (('96->96 @64', 96, 96, 64, 896, ''), ('96->96 @64 +dact', 96, 96, 64, 896, 'dact'),
               ('96->96 @64 +res', 96, 96, 64, 896, 'res'), ('384->384 @16', 384, 384, 16, 896, ''),
               ('pooled 96->192 @64 (ZP 2)', 96, 192, 64, 896, 'pool'), ('up-sampled LN tail 192->96 @32->64 +res (ZP 1)', 192, 96, 64, 896, 'up'))
      if os.environ.get('W4_TRACE_BM64'):               # the 64-cout tile (MF = 2): the reference's default widths
          cases = (('64->64 @64', 64, 64, 64, 960, ''), ('64->64 @64 +dact', 64, 64, 64, 960, 'dact'), ('64->64 @64 +res', 64, 64, 64, 960, 'res'),
                   ('128->128 @32', 128, 128, 32, 960, ''), ('256->256 @16', 256, 256, 16, 960, ''),
                   ('pooled 64->128 @64 (ZP 2)', 64, 128, 64, 960, 'pool'), ('up-sampled LN tail 128->64 @32->64 +res (ZP 1)', 128, 64, 64, 960, 'up'))
      if os.environ.get('W4_TRACE_CASES'):
          cases = [c for c in cases if any(k in c[0] for k in os.environ['W4_TRACE_CASES'].split(','))]
      for name, cin, cout, h, n, epi in cases:
        hs = h // 2 if epi == 'up' else h
        x = torch.randn(n, cin, hs, hs, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        pk = ops.PackedConv(w, None)
        out = torch.empty(n, cout, h // 2, h // 2, device=dev) if epi == 'pool' else torch.empty(n, cout, h, h, device=dev)
        kw = dict(circular=True)
        if epi == 'pool':
            kw.update(pool=(2, 2))
        if epi == 'up':
            kw.update(up=(2, 2), ln=(torch.zeros(n * hs * hs, device=dev), torch.ones(n * hs * hs, device=dev)), res=torch.randn_like(out))
        if epi == 'dact':
            kw.update(dact_z=torch.randn_like(out), act_d=1)
        if epi == 'res':
            kw.update(res=torch.randn_like(out))
        os.environ['SDA_W4_VAR'] = '11'
        for _ in range(12):
            launch_conv(pk, planar_source(x), out, h, h, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch_conv(pk, planar_source(x), out, h, h, **kw)
        e1.record()
        torch.cuda.synchronize()
        wall_ms = e0.elapsed_time(e1)
        os.environ['SDA_W4_VAR'] = '0'
        buf = (ctypes.c_double * 64)()
        lib.sda_w4_trace_read.argtypes = [ctypes.c_void_p]
        rc = lib.sda_w4_trace_read(ctypes.cast(buf, ctypes.c_void_p))
        v = [buf[i] for i in range(64)]
        tiles = n * (h // 8) * (h // 16) * (cout // (96 if cout % 96 == 0 else 64))
        stages = tiles * (cin // 8) / 256
        print(f'--- trace {name}: rc={rc}, ~{stages:.0f} stages per workgroup; cycles per stage by wave and phase; '
              f'launch {wall_ms:.3f} ms -> shader clock {v[7] / (wall_ms * 1e-3) / 1e9:.2f} GHz')
        labels = {0: 'consumer [multiply, M wait, epilogue, E wait]', 4: 'helper [U store, U loads + patch reads, VALU part, M wait, halo loads, E wait]'}
        for wv in range(8):
            print(f'   wave {wv}: ' + ' '.join(f'{v[wv * 8 + k] / stages:8.0f}' for k in range(6)) + f' | total {v[wv * 8 + 7] / stages:8.0f}   ' + labels.get(wv, ''))


def bench():
    print('--- layer bench (algorithmic TFLOP/s; issued = /2.25)')
    for S, n in ((64, 896), (256, 60)):
        layers = [('blk0 96->96 plain', 96, 96, S, {}), ('blk0 96->96 no bias', 96, 96, S, dict(nobias=True)), ('blk0 96->96 mod+LN', 96, 96, S, dict(ln=True, mod=True)),
                  ('blk0 96->96 silu+res', 96, 96, S, dict(silu=True, res=True)), ('blk0^T 96->96 dact', 96, 96, S, dict(dact=True)),
                  ('blk1 192->192 mod+LN', 192, 192, S // 2, dict(ln=True, mod=True)),
                  ('blk2 384->384 mod+LN', 384, 384, S // 4, dict(ln=True, mod=True)),
                  ('tail1 192->96 up+LN', 192, 96, S, dict(ln=True, up=True)), ('tail1 192->96 up+LN+res', 192, 96, S, dict(ln=True, up=True, res=True)), ('tail2 384->192 up+LN', 384, 192, S // 2, dict(ln=True, up=True))]
        for name, cin, cout, h, fz in layers:
            hs = h // 2 if fz.get('up') else h
            x = torch.randn(n, cin, hs, hs, device=dev)
            w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
            b = torch.randn(cout, device=dev)
            pk = ops.PackedConv(w, None if fz.get('nobias') else b)
            out = torch.empty(n, cout, h, h, device=dev)
            kw = dict(circular=True, bias=pk.bias)
            if fz.get('up'):
                kw['up'] = (2, 2)
            if fz.get('ln'):
                kw['ln'] = (torch.zeros(n * hs * hs, device=dev), torch.ones(n * hs * hs, device=dev))
            if fz.get('mod'):
                kw['mod'] = torch.randn(1, cin, device=dev)
            if fz.get('silu'):
                kw['act_in'] = 1
            if fz.get('res'):
                kw['res'] = torch.randn_like(out)
            if fz.get('dact'):
                kw.update(dact_z=torch.randn_like(out), act_d=1)
            res = []
            for use4 in (True, False):
                keep = pk.wino4
                if not use4:
                    pk.wino4 = None
                d = launch_conv(pk, planar_source(x), out, h, h, **kw)
                path = ops.conv_path(d)
                for _ in range(25):                        # (the shader clock takes ~10 launches / 30 ms to ramp up)
                    launch_conv(pk, planar_source(x), out, h, h, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    launch_conv(pk, planar_source(x), out, h, h, **kw)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                pk.wino4 = keep
                flops = 2.0 * n * h * h * cout * cin * 9
                res.append((path, ms, flops / ms / 1e9))
            print(f'S={S:3d} {name:24s} ' + '   '.join(f'path{p}: {ms:7.3f} ms {tf:6.1f} TF (mfma util {tf / 2.25 / 157.3:.2f})' for p, ms, tf in res), flush=True)


if __name__ == '__main__':
    main()
