#!/usr/bin/env python3
"""bench.py -- diffusion-steps/s of the posterior-sampling hot path on MI355X (contract: see the round prompt).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one iteration of VPSDE.sample's loop (sda/score.py:250-261): 1 predictor + C corrector score
evaluations, each = segment-decomposed U-Net score (+ Gaussian-likelihood guidance gradient when --guided 1) over
this rank's batch of trajectories.  Default workload = BASELINE.json configs[3] (the one `metric` is quoted on):
Kolmogorov 64x2x256x256 trajectories, reference Kolmogorov net (kolmogorov/train.py:15-22, 22.9 M params) at size 256,
global batch 128 over 8 GPUs = 16 trajectories per GPU.  Scaling is WEAK: every rank always owns 16 trajectories, so
N=8 is exactly configs[3]; `value` counts whole-job work in units of one diffusion step of a 16-trajectory shard
(value = N*K/T).  Random-init weights, synthetic observations (no datasets/checkpoints are reachable).

Synthetic score net: a raw random-init net makes the *reference itself* overflow under guidance (SURVEY section 0
item 5), so the timed noise estimator is eps(x,t) = sigma x/(mu^2+sigma^2) + 0.1 * UNet(x,t): the exact estimator for
N(0,I) data plus the full U-Net, which still runs (forward and VJP) at every evaluation.

Rank 0 prints ONE JSON line with `roofline` (dominant kernels = the convolutions: Winograd F(2x2,3x3) where eligible,
direct implicit GEMM elsewhere; fp32 MFMA bound; achieved = ALGORITHMIC (direct-convolution) flops / HIP-event time of
the conv launches inside the timed region -- Winograd executes 2.25x fewer multiplies, which is why frac can approach 1) and `cpu_baseline` (the CPU oracle timed on
this box's host cores on a bounded sample of the same workload; N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

K64 = dict(window=5, embedding=64, hidden_channels=(96, 192, 384), hidden_blocks=(3, 3, 3), kernel_size=3, activation='SiLU')

WORKLOADS = {
    # name: (description, kind, net kwargs, event shape, per-GPU batch, obs stride)
    'kolmogorov256': dict(desc='BASELINE configs[3]: Kolmogorov 64x2x256x256, reference Kolmogorov net @256, global batch 128 '
                               'over 8 GPUs = 16 trajectories/GPU', kind='kolmogorov', size=256, L=64, per_gpu=16, state=2),
    'kolmogorov64': dict(desc='BASELINE configs[2]: Kolmogorov 32x2x64x64, reference Kolmogorov net, batch 32 on 1 GPU',
                         kind='kolmogorov', size=64, L=32, per_gpu=32, state=2),
    'qg128': dict(desc='BASELINE configs[4]: QG-shaped 32x4x128x128 (4 state channels), K64-style net, batch 64 over 8 GPUs '
                       '= 8 trajectories/GPU', kind='kolmogorov', size=128, L=32, per_gpu=8, state=4),
    'lorenz63': dict(desc='BASELINE configs[0]: Lorenz-63, L=64, 1-D ScoreUNet (64,)/(3,), batch 1', kind='lorenz', L=64,
                     per_gpu=1, state=3),
    'lorenz96': dict(desc='BASELINE configs[1]: Lorenz-96 40-state, L=128, 1-D ScoreUNet (64,)/(3,), batch 64', kind='lorenz',
                     L=128, per_gpu=64, state=40),
}


class SyntheticScore(torch.nn.Module):
    """eps(x,t) = sigma(t) x / (mu^2 + sigma^2) + scale * net(x, t)   (SURVEY section 8d 'synthetic net').
    mu / sigma are the VP 'cos' schedule of sda/score.py:195-210 (eta = 1e-3)."""

    def __init__(self, net, scale=0.1, eta=1e-3):
        super().__init__()
        self.net, self.scale, self.eta = net, scale, eta

    def forward(self, x, t, c=None):
        sched = getattr(self, '_sched', None)
        if sched is not None:                       # the enclosing VPSDE's schedule (one launch for a device scalar t)
            mu, sigma = sched.mu_sigma(t)
        else:
            mu = torch.cos(math.acos(math.sqrt(self.eta)) * t) ** 2
            sigma = (1 - mu ** 2 + self.eta ** 2).sqrt()
        return x * (sigma / (mu * mu + sigma * sigma)) + self.scale * self.net(x, t, c)


def build_model(wl, device):
    from sda_amd.score import GaussianScore, MCScoreNet, VPSDE
    torch.manual_seed(0)
    if wl['kind'] == 'kolmogorov':
        from sda_amd.experiments.kolmogorov import LocalScoreUNet
        from sda_amd.utils import ACTIVATIONS
        net = MCScoreNet(wl['state'], order=K64['window'] // 2)
        net.kernel = LocalScoreUNet(channels=K64['window'] * wl['state'], size=wl['size'], embedding=K64['embedding'],
                                    hidden_channels=K64['hidden_channels'], hidden_blocks=K64['hidden_blocks'],
                                    kernel_size=3, activation=ACTIVATIONS['SiLU'], spatial=2, padding_mode='circular')
        event = (wl['L'], wl['state'], wl['size'], wl['size'])
        A = lambda x: x[..., ::4, ::4]
    else:
        from sda_amd.experiments.lorenz import make_global_score
        net = make_global_score(channels=wl['state'])
        event = (wl['L'], wl['state'])
        A = lambda x: x[..., ::8, :1]
    return net, event, A


def net_flops_per_window(net_kernel_unet, h, w):
    """Algorithmic conv flops of one U-Net forward on one (h, w) image: 2*MACs over every conv."""
    u = net_kernel_unet
    taps = 1
    for k in u.kernel_size:
        taps *= k
    total, hh, ww = 0.0, h, w
    sizes = []
    for lvl, c in enumerate(u.hidden_channels):
        if lvl > 0:
            hh, ww = (hh + 1) // 2 if u.spatial == 2 else hh, (ww + 1) // 2
        sizes.append((hh, ww))
    for lvl, c in enumerate(u.hidden_channels):
        px = sizes[lvl][0] * sizes[lvl][1]
        cin = u.in_channels if lvl == 0 else u.hidden_channels[lvl - 1]
        total += 2 * taps * px * c * cin                                  # head
        total += 4 * u.hidden_blocks[lvl] * 2 * taps * px * c * c        # 2 convs x (descent + ascent) blocks
        if lvl == 0:
            total += 2 * taps * px * c * u.out_channels                   # tail 0
        else:
            pxu = sizes[lvl - 1][0] * sizes[lvl - 1][1]
            total += 2 * taps * pxu * c * u.hidden_channels[lvl - 1]      # tail after upsample
    return total


def cpu_baseline(wl, args, guided, corrections):
    """The CPU oracle (oracle/sda_oracle.py, pinned against the reference's own code) on a bounded sample of the
    workload: same net family / resolution / guidance, fewer trajectory windows; cost is linear in windows."""
    from oracle import sda_oracle as O
    torch.manual_seed(0)
    ncpu = os.cpu_count() or 1
    if args.cpu_threads:
        cores = min(ncpu, int(args.cpu_threads))
    else:
        # torch's CPU convolutions stop scaling (and then collapse) long before 256 SMT threads on this class of host:
        # pick the fastest of a few thread counts on one representative 3x3 conv, so the baseline is the CPU at its best
        import torch.nn.functional as F
        probe_x, probe_w = torch.randn(4, 96, 64, 64), torch.randn(96, 96, 3, 3)
        best = None
        for th in sorted({t for t in (8, 16, 32, 64, 128) if t <= ncpu} or {ncpu}):
            torch.set_num_threads(th)
            F.conv2d(probe_x, probe_w, padding=1)
            t0 = time.perf_counter()
            for _ in range(5):
                F.conv2d(probe_x, probe_w, padding=1)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, th)
        cores = best[1]
    torch.set_num_threads(cores)
    sched = O.Schedule()
    if wl['kind'] == 'kolmogorov':
        size, state = wl['size'], wl['state']
        order = 2
        cfg = O.UNetConfig(5 * state + 1, 5 * state, 64, (96, 192, 384), (3, 3, 3), 3, 2, 'SiLU', 2, 'circular')
        sd = O.init_score_unet(0, 'kernel.', cfg)
        forcing = O.kolmogorov_forcing(size)
        nwin = max(1, int(args.cpu_windows))
        L = nwin + 2 * order
        x = torch.randn(1, L, state, size, size)
        A = lambda v: v[..., ::4, ::4]

        def net(xx, tt):
            return O.mc_score_net(lambda a, b, c=None: O.score_unet(sd, 'kernel.', cfg, a, b, forcing), order, xx, tt)
        total_windows = wl['per_gpu'] * (wl['L'] - 2 * order)
        unit = f'1 trajectory x {nwin} windows of {state * 5}x{size}x{size}'
    else:
        state = wl['state']
        cfg = O.UNetConfig(state, state, 32, (64,), (3,), 3, 2, 'SiLU', 1, 'zeros')
        sd = O.init_score_unet(0, 'score.', cfg)
        nb = min(wl['per_gpu'], 8)
        x = torch.randn(nb, wl['L'], state)
        A = lambda v: v[..., ::8, :1]

        def net(xx, tt):
            return O.mc_score_wrapper(lambda a, b, c=None: O.score_unet(sd, 'score.', cfg, a, b, c), xx, tt)
        nwin, total_windows = nb, wl['per_gpu']
        unit = f'{nb} trajectories of {wl["L"]}x{state}'

    def eps(xx, tt):
        mu, sg = sched.mu(tt), sched.sigma(tt)
        return xx * (sg / (mu * mu + sg * sg)) + 0.1 * net(xx, tt)

    y = torch.randn(A(x).shape)
    if guided:
        score = lambda xx, tt: O.gaussian_score(eps, sched, y, A, 0.1, 1e-2, xx, tt)
    else:
        def score(xx, tt):
            with torch.no_grad():
                return eps(xx, tt)
    event_ndim = x.dim() - 1
    # one warm step then timed steps until ~args.cpu_seconds
    steps_done, t_spent = 0, 0.0
    xx = x
    time_grid = torch.linspace(1, 0, 1001)
    dt = 1 / 1000

    def one_step(xx, i):
        t = time_grid[i]
        r = sched.mu(t - dt) / sched.mu(t)
        xx = r * xx + (sched.sigma(t - dt) - r * sched.sigma(t)) * score(xx, t)
        for _ in range(corrections):
            z = torch.randn_like(xx)
            e = score(xx, t - dt)
            delta = 0.5 / e.square().mean(dim=tuple(range(1, xx.dim())), keepdim=True)
            xx = xx - (delta * e + torch.sqrt(2 * delta) * z) * sched.sigma(t - dt)
        return xx

    t0 = time.perf_counter()
    xx = one_step(xx, 0)                       # warm-up (thread pool, allocator); also bounds the timed sample
    t_warm = time.perf_counter() - t0
    i = 1
    if t_warm > args.cpu_seconds:              # one step already exceeds the budget: report the warm-up step itself
        steps_done, t_spent = 1, t_warm
    while t_spent < args.cpu_seconds and steps_done < 50:
        t0 = time.perf_counter()
        xx = one_step(xx, i)
        t_spent += time.perf_counter() - t0
        steps_done += 1
        i += 1
    sample_steps_per_s = steps_done / t_spent
    value = sample_steps_per_s * nwin / total_windows        # cost is linear in windows / trajectories
    return dict(value=value, unit='diffusion-steps/s (same per-GPU shard, extrapolated linearly from the sample)',
                cores=cores, kind='port',
                sample=f'{steps_done} timed steps ({t_spent:.1f} s) of {unit}; guided={int(guided)}, corrections={corrections}; '
                       f'scaled by {nwin}/{total_windows}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='kolmogorov256', choices=sorted(WORKLOADS))
    ap.add_argument('--guided', type=int, default=1)
    ap.add_argument('--corrections', type=int, default=1)
    ap.add_argument('--tau', type=float, default=0.5)
    ap.add_argument('--per-gpu', type=int, default=0, help='override trajectories per GPU')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--cpu-windows', type=int, default=2)
    ap.add_argument('--cpu-threads', type=int, default=0, help='0 = calibrate: fastest of 8..128 threads on a probe conv')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--graph', type=int, default=0, help='1: replay each step from a captured hipGraph')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' only to exercise the launch path on a 1-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)            # one rank per GPU on a real node; wraps only in 1-GPU dry runs
    device = torch.device('cuda', dev_index)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)                          # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from sda_amd import ops, parallel
    from sda_amd.score import GaussianScore, VPSDE

    wl = dict(WORKLOADS[args.workload])
    if args.per_gpu:
        wl['per_gpu'] = args.per_gpu
    net, event, A = build_model(wl, device)
    b = wl['per_gpu']
    global_batch = b * world
    score = SyntheticScore(net)
    inner = VPSDE(score, shape=())
    object.__setattr__(score, '_sched', inner)       # plain attribute: not a submodule (inner.eps is score)
    torch.manual_seed(2)
    y = torch.randn(A(torch.empty((b,) + event)).shape)
    eps_mod = GaussianScore(y, A=A, std=0.1, sde=inner) if args.guided else score
    sde = VPSDE(eps_mod, shape=event).to(device)
    # every rank draws its rows of the global noise stream (1-GPU and N-GPU jobs sample the same trajectories)
    sde.initial_noise = parallel.sharded_initial_noise(global_batch, event, 1, rank, world)
    sampler = sde.sampler((b,), steps=1000, corrections=args.corrections, tau=args.tau)

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)           # the RCCL barrier is itself stream work

    if args.graph:
        sampler.capture()
    for _ in range(args.warmup):
        sampler.step()
    sync()
    prof = None
    if rank == 0 and not args.no_profile and not args.graph:
        prof = ops.ConvProfile()
        ops.conv_profile = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sampler.step()
    sync()
    elapsed = time.perf_counter() - t0
    ops.conv_profile = None
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    finite = bool(torch.isfinite(sampler.x).all().item())
    # the one collective of the job: gather the samples (after the loop; not part of a step)
    t1 = time.perf_counter()
    gathered = parallel.all_gather_samples(sampler.x, global_batch)
    sync()
    gather_ms = (time.perf_counter() - t1) * 1e3
    assert gathered.shape[0] == global_batch

    if rank == 0:
        value = world * args.steps / elapsed
        out = {
            'metric': 'diffusion-steps/s (Kolmogorov 64x256x256 posterior sampling; unit = one predictor-corrector step of a '
                      '16-trajectory per-GPU shard)' if args.workload == 'kolmogorov256' else f'diffusion-steps/s ({args.workload})',
            'value': value, 'unit': 'diffusion-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic (random-init net + exact Gaussian term, synthetic observations)',
            'config': {'workload': args.workload, 'description': wl['desc'], 'event': list(event),
                       'per_gpu_batch': b, 'global_batch': global_batch, 'guided': bool(args.guided),
                       'corrections': args.corrections, 'tau': args.tau, 'schedule_steps': 1000,
                       'score_evals_per_step': 1 + args.corrections, 'hipgraph_step': bool(args.graph), 'parallelism': f'dp{world} (batch-sharded, no in-loop collective)'},
            'wallclock_per_1000_steps_s': elapsed / args.steps * 1000,
            'samples_finite': finite, 'final_allgather_ms': gather_ms,
        }
        if prof is not None:
            s = prof.summary()
            ach = s['total_flops'] / (s['total_ms'] * 1e-3) / 1e12 if s['total_ms'] > 0 else 0.0
            traffic = None
            tfile = os.path.join(ROOT, 'profiles', f'r01_{args.workload}_g{int(bool(args.guided))}c{args.corrections}_traffic.json')
            if os.path.exists(tfile):          # HBM bytes per conv launch from the committed rocprofv3 PMC passes of this command
                traffic = json.load(open(tfile)).get('hbm_bytes_per_launch')
            out['roofline'] = {'bound': 'mfma', 'kernel': 'conv_wino_kernel + conv_igemm_ws_kernel (fp32 v_mfma_f32_32x32x2_f32; all convolution launches)', 'achieved': ach,
                               'peak': 157.3, 'unit': 'TFLOP/s', 'frac': ach / 157.3, 'traffic': traffic,
                               'traffic_source': os.path.basename(tfile) if traffic is not None else None,
                               'launches': s['launches'], 'avg_launch_ms': s['total_ms'] / max(1, s['launches']),
                               'avg_launch_gflop': s['total_flops'] / max(1, s['launches']) / 1e9,
                               'conv_time_share_of_step': s['total_ms'] * 1e-3 / elapsed}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(wl, args, bool(args.guided), args.corrections)
            out['gpu_over_cpu'] = value / out['cpu_baseline']['value']
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
