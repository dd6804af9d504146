#!/usr/bin/env python3
"""Find the first non-finite convolution output of a bench-shaped f16x2 step (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sda_amd import engine, ops, parallel
from sda_amd.score import GaussianScore, VPSDE

ops.set_multiply('f16x2')
dev = torch.device('cuda:0')
wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'kolmogorov64'])
wl['per_gpu'] = int(sys.argv[2]) if len(sys.argv) > 2 else 2
net, event, A = bench.build_model(wl, dev)
score = bench.SyntheticScore(net)
inner = VPSDE(score, shape=())
object.__setattr__(score, '_sched', inner)
y, x_init = bench.rank_inputs(wl, event, 'weak', 0, 1)
sde = VPSDE(GaussianScore(y, A=A, std=0.1, sde=inner), shape=event).to(dev)
sde.initial_noise = x_init
sde.noise_source = parallel.KeyedNoise((0, wl['per_gpu']), event, 2, 1, dev)
real = engine.launch_conv
count = [0]


def checked(pk, src, out, ho, wo, **kw):
    d = real(pk, src, out, ho, wo, **kw)
    count[0] += 1
    if d is not None and not torch.isfinite(out).all():
        xa = kw.get('x_amax')
        print(f'launch {count[0]}: non-finite output; h2={bool(d.w_h2)} cx={d.cx} cout={d.cout} hw={d.ho}x{d.wo} n={d.n} ln={kw.get("ln") is not None} act_in={kw.get("act_in", 0)} '
              f'dact={kw.get("dact_z") is not None} res={kw.get("res") is not None} x_amax={None if xa is None else (float(xa) if torch.is_tensor(xa) else xa)} '
              f'w_scale={getattr(pk, "h2_scale", None)}')
        for k in ('res', 'dact_z'):
            if kw.get(k) is not None:
                print('   ', k, 'finite:', bool(torch.isfinite(kw[k]).all()))
        raise SystemExit(1)
    return d


engine.launch_conv = checked
sampler = sde.sampler((wl['per_gpu'],), steps=1000, corrections=1, tau=0.5)
if len(sys.argv) > 3 and sys.argv[3] == 'graph':
    engine.launch_conv = real
    sampler.capture()
for i in range(int(os.environ.get("H2_STEPS", "3"))):
    sampler.step()
    if os.environ.get('H2_PROBE') and i == int(os.environ['H2_PROBE']):
        mode = os.environ.get('H2_PROBE_MODE', 'probe')
        if mode == 'probe':
            print('probe', ops.clock_probe(dev)['ghz'])
        elif mode == 'alloc':
            t = torch.zeros(2048, device=dev, dtype=torch.int64); torch.cuda.synchronize(); print('alloc only', t.data_ptr())
        elif mode == 'sync':
            torch.cuda.synchronize(); print('sync only')
        elif mode == 'matmul':
            a_ = torch.randn(4096, 4096, device=dev); print('matmul', float((a_ @ a_).sum()))
    print('step', i, 'finite', bool(torch.isfinite(sampler.x).all()), 'max |x|', float(sampler.x.abs().max()), 'launches', count[0])
