#!/usr/bin/env python3
"""Randomised parity sweep of sda_conv_igemm (direct, Winograd, parity-class and fallback kernels) against torch fp64.

    python tests/fuzz/conv_fuzz.py [--cases 300] [--seed 0]

Every case draws a layer shape and a random subset of the loader / epilogue fusions, runs the HIP path and compares with
a float64 torch restatement at 1e-4 scale-relative (the north_star tolerance).  Prints failures with their configuration."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sda_amd import ops  # noqa: E402
from sda_amd._lib import ACT_IDS  # noqa: E402
from sda_amd.engine import launch_conv, planar_source  # noqa: E402

ACTS = {'SiLU': F.silu, 'GELU': F.gelu, 'ELU': F.elu, 'ReLU': torch.relu, 'SELU': F.selu}


def ref_conv(x, w, b, stride, circular, kh, kw):
    ph, pw = kh // 2, kw // 2
    xp = F.pad(x, (pw, pw, ph, ph), mode='circular' if circular else 'constant')
    return F.conv2d(xp, w, b, stride=stride)


def one_case(rng, dev, idx, large=False):
    """large=True: images of 128 / 256 pixels per side (the configs[3] / [4] U-Net levels), one image, fewer channels."""
    mode = rng.choice(['plain', 'plain', 'wino', 'wino', 'wino4', 'wino4', 'stride2', 'up', 'zins', 'oned'])
    if large and mode == 'oned':
        mode = 'wino4'
    circular = rng.random() < 0.6
    cfg = dict(mode=mode, circular=circular)
    if mode == 'wino':
        cin = rng.choice([8, 24, 96, 100, 192])
        cout = rng.choice([96, 192])
        h, w_ = rng.choice([2, 4, 6, 8, 16, 18, 32, 64]), rng.choice([2, 4, 8, 10, 16, 32, 64])
        if large:
            cin, cout = rng.choice([8, 24, 96]), 96
            h, w_ = rng.choice([128, 256]), rng.choice([128, 256])
        kh = kw = 3
    elif mode == 'wino4':
        # shapes the one-wave-per-SIMD Winograd kernel tiles (height % 8 == 0, width % 16 == 0, cout % 96 == 0 -- or, on its 64-cout
        # tile (round 6), cout % 64 == 0: the reference's default widths (64, 128, 256)), incl. partial last K-stages (cin % 16 != 0),
        # the upsampling tails and both paddings
        cin = rng.choice([3, 8, 16, 24, 40, 56, 64, 96, 100, 128, 192])
        cout = rng.choice([96, 96, 192, 64, 64, 128, 256, 32])
        h, w_ = rng.choice([8, 16, 24, 32, 64]), rng.choice([16, 32, 48, 64])
        if large:
            cin, cout = rng.choice([16, 24, 96]), 96
            h, w_ = rng.choice([128, 256]), rng.choice([128, 256])
        kh = kw = 3
    elif mode == 'oned':
        cin, cout = rng.choice([3, 5, 64, 70]), rng.choice([3, 64, 96, 130])
        h, w_ = 1, rng.choice([1, 2, 7, 16, 65, 128, 300])
        kh, kw = 1, 3
    else:
        cin, cout = rng.choice([1, 2, 7, 11, 24, 96, 97]), rng.choice([1, 3, 10, 32, 33, 96, 128, 160])
        h, w_ = rng.choice([1, 2, 3, 5, 8, 16, 31, 64]), rng.choice([1, 2, 4, 5, 9, 16, 33, 64])
        if large:
            cin, cout = rng.choice([2, 11, 21, 96]), rng.choice([10, 20, 33, 96])
            h, w_ = rng.choice([128, 256]), rng.choice([128, 256])
            if mode in ('up', 'zins'):
                h, w_ = h // 2, w_ // 2
        kh = kw = 3
    n = rng.choice([1, 2, 3, 5])
    if cin * h * w_ * n > 3e6:
        n = 1
    cfg.update(n=n, cin=cin, cout=cout, h=h, w=w_)
    g = torch.Generator().manual_seed(1000 + idx)
    x = torch.randn(n, cin, h, w_, generator=g) * 1.5 + 0.2
    wgt = torch.randn(cout, cin, kh, kw, generator=g) / (kh * kw * cin) ** 0.5
    bias = torch.randn(cout, generator=g) if rng.random() < 0.7 else None
    opts, xin = {}, x.double()
    # loader fusions
    use_mod = rng.random() < 0.5
    use_ln = rng.random() < 0.5 and cin > 1
    per_image = rng.random() < 0.5
    act = rng.choice([None, None, 'SiLU', 'GELU', 'ELU', 'ReLU', 'SELU'])
    if mode == 'wino4':
        act = rng.choice([None, 'SiLU', 'SiLU', 'GELU'])          # (GELU: falls back to the first-generation kernel)
        per_image = False                                         # (per-image modulation: direct kernel)
    cfg.update(mod=use_mod, ln=use_ln, per_image=per_image, act=act, bias=bias is not None)
    if use_mod:
        mod = torch.randn(n if per_image else 1, cin, generator=g)
        opts['mod'], opts['mod_sn'] = mod, (cin if per_image else 0)
        xin = xin + mod.double()[:, :, None, None]
    if use_ln:
        var, mean = torch.var_mean(xin, dim=1, unbiased=True, keepdim=True)
        rstd = 1 / torch.sqrt(var + 1e-5)
        opts['ln'] = (mean.float().reshape(n, -1), rstd.float().reshape(n, -1))
        xin = (xin - mean) * rstd
    if act:
        opts['act_in'] = ACT_IDS[act]
        xin = ACTS[act](xin)
    stride = (1, 1)
    ho, wo = h, w_
    if mode == 'stride2':
        stride = (2, 2)
        ho, wo = (h - 1) // 2 + 1, (w_ - 1) // 2 + 1
        if circular and (h % 2 or w_ % 2):
            circular = cfg['circular'] = False
        opts['stride'] = stride
        ref = ref_conv(xin, wgt.double(), None if bias is None else bias.double(), 2, circular, kh, kw)
    elif mode == 'wino4' and rng.random() < 0.3:
        cfg['up'] = True
        opts['up'] = (2, 2)
        h2, w2 = h // 2, w_ // 2
        x = x[:, :, :h2, :w2].contiguous()
        xin = xin[:, :, :h2, :w2]
        if use_ln:
            opts['ln'] = tuple(t.reshape(n, h, w_)[:, :h2, :w2].reshape(n, -1).contiguous() for t in opts['ln'])
        xin = xin.repeat_interleave(2, -1).repeat_interleave(2, -2)
        ref = ref_conv(xin, wgt.double(), None if bias is None else bias.double(), 1, circular, kh, kw)
    elif mode == 'up':
        uh = 1 if h == 1 else 2
        opts['up'] = (uh, 2)
        ho, wo = h * uh, w_ * 2
        xin = xin.repeat_interleave(2, -1)
        if uh == 2:
            xin = xin.repeat_interleave(2, -2)
        ref = ref_conv(xin, wgt.double(), None if bias is None else bias.double(), 1, circular, kh, kw)
    elif mode == 'zins':
        # backward-data of a stride-2 conv: x plays the gradient at (h, w); output is (2h, 2w)
        ho, wo = 2 * h, 2 * w_
        xx = torch.zeros(n, cout, ho, wo, dtype=torch.float64, requires_grad=True)
        y = ref_conv(xx, wgt.transpose(0, 1).contiguous().double(), None, 2, circular, kh, kw)   # weight (cin_fwd_out=cin.., )
        ref, = torch.autograd.grad(y, xx, xin)
        bias = None
        cfg['bias'] = False
    else:
        ref = ref_conv(xin, wgt.double(), None if bias is None else bias.double(), 1, circular, kh, kw)
    # epilogue fusions
    if rng.random() < 0.4:
        actd = rng.choice(list(ACTS))
        z = torch.randn(ref.shape, generator=g)
        zz = z.double().requires_grad_(True)
        dz, = torch.autograd.grad(ACTS[actd](zz).sum(), zz)
        ref = ref * dz
        opts['dact_z'], opts['act_d'] = z, ACT_IDS[actd]
        cfg['dact'] = actd
    if rng.random() < 0.5:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
        opts['res'] = res
        cfg['res'] = True
    # HIP path
    xd = x.to(dev)
    if mode == 'zins':
        pk = ops.PackedConv(wgt.transpose(0, 1).contiguous().to(dev), None, transpose=True)
        opts['zins'] = (2, 2)
    else:
        pk = ops.PackedConv(wgt.to(dev), None if bias is None else bias.to(dev))
    out = torch.full((n, pk.m_real, ho, wo), float('nan'), device=dev)
    dopts = {}
    for k, v in opts.items():
        if torch.is_tensor(v):
            dopts[k] = v.to(dev).contiguous()
        elif k == 'ln':
            dopts[k] = tuple(t.to(dev).contiguous() for t in v)
        else:
            dopts[k] = v
    try:
        desc = launch_conv(pk, planar_source(xd), out, ho, wo, circular=circular, bias=pk.bias, **dopts)
    except Exception as e:  # noqa: BLE001
        return cfg, f'EXCEPTION {type(e).__name__}: {e}'
    cfg['path'] = ops.conv_path(desc)
    # the second-generation Winograd kernel serves the four loader configurations of the reference U-Net
    w4_cfg = (use_mod, use_ln, act == 'SiLU') in ((False, False, False), (False, False, True), (False, True, False), (True, True, False))
    # (5 = its zero-position form)
    if mode == 'wino4' and act in (None, 'SiLU') and w4_cfg and ops.WINOGRAD4 and ops.WINOGRAD and (cout % 96 == 0 or ops.WINO4_BM64) and cfg['path'] not in (2, 5):
        return cfg, f'expected the second-generation Winograd kernel, got path {cfg["path"]}'
    torch.cuda.synchronize()
    got = out.cpu().double()
    if torch.isnan(got).any():
        return cfg, 'NaN in output (unwritten elements)'
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    if err > 1e-4 * scale + 1e-6:
        return cfg, f'max abs err {err:.3e} vs scale {scale:.3e}'
    return cfg, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=300)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--large', action='store_true', help='128 / 256-pixel images')
    args = ap.parse_args()
    rng = random.Random(args.seed)
    dev = torch.device('cuda:0')
    bad = 0
    modes = {}
    for i in range(args.cases):
        try:
            cfg, msg = one_case(rng, dev, i + 7919 * args.seed, large=args.large)
        except Exception as e:  # noqa: BLE001
            cfg, msg = {'case': i}, f'EXCEPTION {type(e).__name__}: {e}'
        modes[cfg.get('mode')] = modes.get(cfg.get('mode'), 0) + 1
        if msg:
            bad += 1
            print(f'FAIL case {i}: {msg}\n     {cfg}', flush=True)
    print(f'{args.cases - bad}/{args.cases} cases within 1e-4; by mode: {modes}')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
