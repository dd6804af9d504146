#!/usr/bin/env python3
"""bench.py -- diffusion-steps/s of the posterior-sampling hot path on MI355X (contract: see the round prompt).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under torch.distributed.run)
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

Rank 0 prints ONE JSON line with `roofline` and `cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded
sample of the same workload; N=1 only).

roofline: the dominant kernel is the Winograd F(2x2,3x3) convolution (conv_wino4_kernel; ~85 % of a step).  It is
MFMA-bound, and what it ISSUES is 1/2.25 of the algorithmic (direct-convolution) flops, so
    achieved = issued fp32 MFMA TFLOP/s = algorithmic flops of its launches / 2.25 / their HIP-event time,
    frac     = achieved / 157.3 TFLOP/s  = the matrix-pipe utilisation (what rocprof's SQ_INSTS_MFMA x 512 flop / time gives),
with the direct-equivalent figure, the other kernel families (direct implicit GEMM: TFLOP/s vs the same peak; LayerNorm:
GB/s vs 8 TB/s) and their shares of the step alongside in `roofline.families`.  The step is replayed from a captured hipGraph
in the timed region (`--graph 1`, default); the per-kernel HIP-event timings come from `--profile-steps` extra EAGER steps run
after it (events cannot bracket launches inside a graph replay) -- same kernels, same shapes, outside `value`.
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
    # the reference's own DEFAULT widths (experiments/kolmogorov/utils.py:52 make_score(): window 3, (64, 128, 256)) -- none a multiple of 96
    'kolmogorov64_default': dict(desc='make_score() defaults of experiments/kolmogorov/utils.py:49-57 (window 3, hidden_channels (64, 128, 256), '
                                      'blocks (3, 3, 3)): Kolmogorov 32x2x64x64, batch 32 on 1 GPU', kind='kolmogorov', size=64, L=32, per_gpu=32,
                                 state=2, net=dict(window=3, embedding=64, hidden_channels=(64, 128, 256), hidden_blocks=(3, 3, 3))),
    'qg128': dict(desc='BASELINE configs[4]: QG-shaped 32x4x128x128 (4 state channels), K64-style net, batch 64 over 8 GPUs '
                       '= 8 trajectories/GPU', kind='kolmogorov', size=128, L=32, per_gpu=8, state=4),
    'lorenz63': dict(desc='BASELINE configs[0]: Lorenz-63, L=64, 1-D ScoreUNet (64,)/(3,), batch 1', kind='lorenz', L=64,
                     per_gpu=1, state=3),
    'lorenz96': dict(desc='BASELINE configs[1]: Lorenz-96 40-state, L=128, 1-D ScoreUNet (64,)/(3,), batch 64', kind='lorenz',
                     L=128, per_gpu=64, state=40),
    # the reference's canonical caller (experiments/lorenz/eval.py:72-84), run as a whole: see run_lorenz_eval
    'lorenz_eval': dict(desc='experiments/lorenz/eval.py:72-84: sde.sample((1024,), steps=256, corrections=C, tau=0.25) for C in '
                             '(0, 1, 2, 4, 8, 16), event (65, 3), GaussianScore(A = x[..., ::step, :1], std, gamma=3e-2)',
                        kind='lorenz_eval', L=65, per_gpu=1024, state=3),
}
LORENZ_EVAL_FREQ = {'lo': (8, 0.05), 'hi': (1, 0.25)}          # (step, std): experiments/lorenz/eval.py:50-53


class SyntheticScore(torch.nn.Module):
    """eps(x,t) = sigma(t) x / (mu^2 + sigma^2) + scale * net(x, t)   (SURVEY section 8d 'synthetic net'; mu^2 + sigma^2 = 1 + eta^2).
    mu / sigma are the VP 'cos' schedule of sda/score.py:195-210 (eta = 1e-3)."""

    def __init__(self, net, scale=0.1, eta=1e-3):
        super().__init__()
        self.net, self.scale, self.eta = net, scale, eta

    def affine_form(self, sched):
        """(net, cx0, cx1, cn) with eps(x, t) = (cx0 + cx1 sigma(t)) x + cn net(x, t), sigma = `sched`'s -- the protocol by which
        sda_amd's fused 1-D evaluation (sda_amd/fused1d.py) takes an estimator of this form; None: not expressible (another
        schedule than the one forward() uses)."""
        if getattr(self, '_sched', None) is not sched:
            return None
        return self.net, 0.0, 1.0 / (1.0 + self.eta ** 2), self.scale

    def forward(self, x, t, c=None):
        sched = getattr(self, '_sched', None)
        if sched is not None:                       # the enclosing VPSDE's schedule (one launch for a device scalar t)
            mu, sigma = sched.mu_sigma(t)
        else:
            mu = torch.cos(math.acos(math.sqrt(self.eta)) * t) ** 2
            sigma = (1 - mu ** 2 + self.eta ** 2).sqrt()
        # (sigma^2 = 1 - mu^2 + eta^2 for every VP schedule (sda/score.py:209-210), so mu^2 + sigma^2 = 1 + eta^2: the coefficient is one
        # scalar launch, the expression three elementwise launches instead of seven, two in its autograd VJP: on the latency-bound
        # Lorenz workloads every scalar-tensor launch is ~1 % of a step)
        k = sigma * (1.0 / (1.0 + self.eta ** 2))
        return torch.add(x * k, self.net(x, t, c), alpha=self.scale)


def build_model(wl, device):
    from sda_amd import observe as Ob
    from sda_amd.score import GaussianScore, MCScoreNet, VPSDE
    torch.manual_seed(0)
    if wl['kind'] == 'kolmogorov':
        from sda_amd.experiments.kolmogorov import LocalScoreUNet
        from sda_amd.utils import ACTIVATIONS
        nk = {**K64, **wl.get('net', {})}
        net = MCScoreNet(wl['state'], order=nk['window'] // 2)
        net.kernel = LocalScoreUNet(channels=nk['window'] * wl['state'], size=wl['size'], embedding=nk['embedding'],
                                    hidden_channels=nk['hidden_channels'], hidden_blocks=nk['hidden_blocks'],
                                    kernel_size=3, activation=ACTIVATIONS['SiLU'], spatial=2, padding_mode='circular')
        event = (wl['L'], wl['state'], wl['size'], wl['size'])
        A = Ob.Subsample.space(4)                    # x[..., ::4, ::4] with a hand-written adjoint: no autograd through A
    else:
        from sda_amd.experiments.lorenz import make_global_score
        net = make_global_score(channels=wl['state'])
        event = (wl['L'], wl['state'])
        A = Ob.Subsample((slice(None, None, 8), slice(0, 1)))        # x[..., ::8, :1]
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
    sched = O.Schedule()
    if wl['kind'] == 'kolmogorov':
        size, state = wl['size'], wl['state']
        nk = {**K64, **wl.get('net', {})}
        order = nk['window'] // 2
        cfg = O.UNetConfig(nk['window'] * state + 1, nk['window'] * state, nk['embedding'], tuple(nk['hidden_channels']),
                           tuple(nk['hidden_blocks']), 3, 2, 'SiLU', 2, 'circular')
        sd = O.init_score_unet(0, 'kernel.', cfg)
        forcing = O.kolmogorov_forcing(size)
        nwin = max(1, int(args.cpu_windows))
        L = nwin + 2 * order
        x = torch.randn(1, L, state, size, size)
        A = lambda v: v[..., ::4, ::4]

        def net(xx, tt):
            return O.mc_score_net(lambda a, b, c=None: O.score_unet(sd, 'kernel.', cfg, a, b, forcing), order, xx, tt)
        total_windows = wl['per_gpu'] * (wl['L'] - 2 * order)
        unit = f'1 trajectory x {nwin} windows of {state * nk["window"]}x{size}x{size}'
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

    # thread count: the fastest of a few candidates ON THE WORKLOAD ITSELF (one score evaluation of the sample; torch's CPU
    # convolutions stop scaling long before the SMT thread count of this class of host, and a small probe convolution does
    # not predict where) -- the baseline is the CPU at its best.  The probe evaluations double as the thread-pool warm-up.
    if args.cpu_threads:
        cores, probe = min(ncpu, int(args.cpu_threads)), 'given'
    else:
        cands = sorted({t for t in (16, 32, 64, 128) if t <= ncpu} or {ncpu})
        best = None
        for th in cands:
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            score(x, time_grid[0])
            dtp = time.perf_counter() - t0
            if best is None or dtp < best[0]:
                best = (dtp, th)
        cores, probe = best[1], 'fastest of ' + '/'.join(map(str, cands)) + ' on one score evaluation of the sample'
    torch.set_num_threads(cores)
    # >= 1 untimed warm-up step, then >= 3 timed steps, more while the budget lasts
    xx = one_step(x, 0)
    steps_done, t_spent, i, per_step = 0, 0.0, 1, []
    while steps_done < 3 or (t_spent < args.cpu_seconds and steps_done < 50):
        t0 = time.perf_counter()
        xx = one_step(xx, i)
        per_step.append(time.perf_counter() - t0)
        t_spent += per_step[-1]
        steps_done += 1
        i += 1
    per_step.sort()
    # the MEDIAN timed step (round 6; the mean had drifted +-15 % between rounds with the host's other tenants: VERDICT r5 weak 8)
    sample_steps_per_s = 1.0 / per_step[len(per_step) // 2]
    value = sample_steps_per_s * nwin / total_windows        # cost is linear in windows / trajectories
    return dict(value=value, unit='diffusion-steps/s (same per-GPU shard, extrapolated linearly from the sample)',
                cores=cores, kind='port', cpu_model=_cpu_model(), host_logical_cpus=ncpu, threads_chosen_by=probe,
                # (how soft the figure is: the spread of the timed steps themselves; box to box it has been +-20 %)
                sample_step_s={'min': per_step[0], 'median': per_step[len(per_step) // 2], 'max': per_step[-1]},
                value_from='median timed step', mean_based_value=steps_done / t_spent * nwin / total_windows,
                sample=f'{steps_done} timed steps after 1 warm-up step ({t_spent:.1f} s, {t_spent / steps_done:.2f} s/step) of {unit}; '
                       f'guided={int(guided)}, corrections={corrections}; scaled by {nwin}/{total_windows}')


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


PEAK_MFMA_F32 = 157.3      # TFLOP/s, dense fp32 MFMA (MI355X_MICROARCH.md)
PEAK_MFMA_F16 = 2500.0     # TFLOP/s, dense f16 / bf16 MFMA (same guide; the 5 PFLOP/s headline figure includes 2:1 sparsity)
PEAK_HBM = 8000.0          # GB/s


def roofline_report(prof, prof_steps, step_s, args, root, clock_probe=None):
    """Per kernel family, from HIP-event timings of `prof_steps` steps.  The dominant family's ISSUED MFMA rate over the fp32
    matrix peak is `frac` (Winograd F(2x2,3x3) issues 1/2.25 of the algorithmic flops); the direct-equivalent rate is an extra."""
    s = prof.summary()
    fam = {}
    for name, r in s['families'].items():
        alg = r['flops'] / (r['ms'] * 1e-3) / 1e12 if r['ms'] > 0 else 0.0
        # (zero-position form of conv_wino4: 54 of the 96 multiplies of a Winograd stage are issued)
        # (f16 x 2 direct convolution: three f16 MFMA products per multiply, priced against the f16 matrix peak)
        issued = alg / 4.0 if name == 'wino4zp' else alg / 2.25 if name.startswith('wino') else 3.0 * alg if name in ('h2', 'h2s2') else 3.0 * alg * 4 / 9 if name == 'h2up' else alg
        fam[name] = {'launches_per_step': r['launches'] / prof_steps, 'ms_per_step': r['ms'] / prof_steps,
                     'share_of_step': r['ms'] * 1e-3 / prof_steps / step_s,
                     'algorithmic_tflops': alg, 'issued_mfma_tflops': issued, 'mfma_util': issued / (PEAK_MFMA_F16 if name in ('h2', 'h2up', 'h2s2') else PEAK_MFMA_F32),
                     'avg_launch_ms': r['ms'] / max(1, r['launches']),
                     'avg_launch_gflop_algorithmic': r['flops'] / max(1, r['launches']) / 1e9}
        if name in prof.family_bytes:
            rd, wr = prof.family_bytes[name]
            fam[name]['avg_launch_algorithmic_read_GB'] = rd / max(1, r['launches']) / 1e9
            fam[name]['avg_launch_algorithmic_write_GB'] = wr / max(1, r['launches']) / 1e9
    for name, r in prof.mem_summary().items():
        gbs = r['bytes'] / (r['ms'] * 1e-3) / 1e9 if r['ms'] > 0 else 0.0
        fam[name] = {'launches_per_step': r['launches'] / prof_steps, 'ms_per_step': r['ms'] / prof_steps,
                     'share_of_step': r['ms'] * 1e-3 / prof_steps / step_s, 'bound': 'hbm',
                     'algorithmic_GBps': gbs, 'frac_of_8TBps': gbs / PEAK_HBM}
    conv = {k: v for k, v in fam.items() if 'issued_mfma_tflops' in v}
    dom = max(conv, key=lambda k: conv[k]['ms_per_step']) if conv else None
    traffic, tfile, tracked = None, None, {}
    suffix = '' if args.multiply == 'f32' else '_' + args.multiply
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):      # HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes
        tfile = os.path.join(root, 'profiles', f'{rnd}_{args.workload}_g{int(bool(args.guided))}c{args.corrections}{suffix}_traffic.json')
        if os.path.exists(tfile):
            tracked = json.load(open(tfile))
            traffic = tracked.get('hbm_bytes_per_launch')
            break
    kernel = {'h2': 'conv_h2_kernel (direct 3x3, fp32 multiply emulated as three f16 products, v_mfma_f32_32x32x16_f16)',
              'h2s2': 'conv_h2_kernel, stride-2 forms (heads over input parity planes, their VJP as four output parity classes: 1 / 2 / 2 / 4 taps)',
              'h2up': 'conv_h2_kernel, up-sampled form (four output parity classes of 2 x 2 pre-summed taps: 4 of 9 taps issued)',
              'wino4': 'conv_wino4_kernel (Winograd F(2x2,3x3), v_mfma_f32_16x16x4_f32)',
              'wino4zp': 'conv_wino4_kernel, zero-position form (up-sampling tails and their VJP: 9 of 16 Winograd positions)',
              'wino': 'conv_wino_kernel (Winograd F(2x2,3x3), v_mfma_f32_32x32x2_f32)',
              'direct': 'conv_igemm_ws_kernel (direct implicit GEMM, v_mfma_f32_32x32x2_f32)',
              'par4': 'conv_par4_kernel (stride-2 backward-data, four parity classes in one launch, v_mfma_f32_32x32x2_f32)',
              'few': 'conv_few_kernel (3x3, <= 16 output channels, v_mfma_f32_16x16x4_f32)',
              'small1d': 'conv_small1d_kernel (single-round-trip 1-D convolution, v_mfma_f32_16x16x4_f32)',
              'net1d_fwd': 'net1d_fwd_kernel (whole single-level 1-D U-Net in one launch, v_mfma_f32_16x16x4_f32)',
              'net1d_bwd': 'net1d_bwd_kernel (its input VJP in one launch, v_mfma_f32_16x16x4_f32)',
              'block1d_fwd': 'block1d_fwd_kernel (fused 1-D residual block, v_mfma_f32_16x16x4_f32)',
              'block1d_bwd': 'block1d_bwd_kernel (fused 1-D residual block VJP, v_mfma_f32_16x16x4_f32)'}.get(dom, dom)
    d = conv.get(dom, {})
    is_h2 = dom in ('h2', 'h2up', 'h2s2')             # (one predicate for peak, frac_algorithmic and the note: ADVICE r5)
    # the clock `frac` was measured at.  The peak is quoted at 2.4 GHz; the boxes of the pool sustain 2.17-2.29 GHz under this kernel
    # (power-limited, data dependent), so the same build reads 0.70-0.73 depending on the box.  Two readings: the matrix-core
    # stream's own clock on THIS box (sda_clock_probe, run after the warm-up steps), and the dominant kernel's clock = its cycles per
    # launch (GRBM_GUI_ACTIVE / 8 XCDs from the tracked PMC pass of this build -- a property of the code) / its live launch time.
    clock = None
    if clock_probe is not None and d.get('mfma_util') is not None:
        clock = {'peak_quoted_at_GHz': 2.4, 'probe': clock_probe, 'frac_at_probe_clock': d['mfma_util'] * 2.4 / clock_probe['ghz']}
        cyc = (tracked.get('dominant_kernel_counters_per_launch') or {}).get('GRBM_GUI_ACTIVE')
        if cyc and dom == 'wino4' and d.get('avg_launch_ms'):
            ghz = cyc / 8 / (d['avg_launch_ms'] * 1e-3) / 1e9
            clock.update(dominant_kernel_GHz=ghz, dominant_kernel_cycles_per_launch=cyc / 8,
                         frac_of_kernel_cycles=d['mfma_util'] * 2.4 / ghz,
                         cycles_source=os.path.basename(tfile) + ' (GRBM_GUI_ACTIVE / 8, tracked PMC pass)')
    latency = None
    if dom in ('small1d', 'block1d_fwd', 'block1d_bwd', 'net1d_fwd', 'net1d_bwd'):
        # the 1-D nets are LATENCY-bound (a launch is a few hundred kFLOP per image): the model that prices them is launches per
        # step x time per launch against the floor of a dependent launch in a graph chain (1.7 us measured), not a
        # fraction of the matrix peak.  HIP-event brackets add their own few us per launch: `event_us_per_launch` is an upper
        # bound, `graph_us_per_launch_all_kernels` = the graph-replayed step / every launch of the step is the clean figure.
        n_launch = sum(v['launches_per_step'] for v in fam.values())
        latency = {'model': 'launches/step x us/launch vs the 1.7 us dependent-launch floor',
                   'bracketed_launches_per_step': n_launch,
                   'per_family': {k: {'launches_per_step': v['launches_per_step'], 'event_us_per_launch': 1e3 * v['avg_launch_ms']}
                                  for k, v in fam.items()},
                   'step_us': step_s * 1e6, 'launch_floor_us': 1.7,
                   'floor_us_per_step_bracketed_kernels': 1.7 * n_launch}
    return {'bound': 'latency' if latency else 'mfma', 'latency': latency,
            'kernel': kernel, 'achieved': d.get('issued_mfma_tflops'), 'peak': PEAK_MFMA_F16 if is_h2 else PEAK_MFMA_F32, 'unit': 'TFLOP/s',
            'frac': d.get('mfma_util'),
            # SURVEY 8(d)'s definition (ALGORITHMIC flops of the direct convolution / time / peak) next to the issued one: above 1 for a
            # Winograd kernel (it issues 1 / 2.25 of them), a third of `frac`'s numerator for the three-product f16 x 2 kernel
            'frac_algorithmic': None if d.get('algorithmic_tflops') is None else d['algorithmic_tflops'] / (PEAK_MFMA_F16 if is_h2 else PEAK_MFMA_F32),
            'sustained_note': ('tools/h2_power_probe.hip (profiles/r05_h2_power_probe.txt): on full-mantissa random halves this instruction mix '
                               'sustains 1 592 TFLOP/s = 0.64 of the quoted peak (power-managed clock 1.64 GHz; 2 411 on zeros)') if is_h2 else None,
            'clock': clock, 'traffic': traffic, 'traffic_source': os.path.basename(tfile) if traffic is not None else None,
            'achieved_is': 'ISSUED fp32 MFMA flops of the dominant kernel (algorithmic / 2.25 for Winograd) / its HIP-event time',
            'traffic_is': 'HBM/fabric bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes: 2 x FETCH_SIZE + WRITE_SIZE '
                          '(calibrated on known byte counts in the kernel\'s own access shapes, tools/fetch_calib.hip); compare with '
                          'algorithmic_bytes_per_launch',
            'algorithmic_bytes_per_launch': None if d.get('avg_launch_algorithmic_read_GB') is None else
            1e9 * (d['avg_launch_algorithmic_read_GB'] + d['avg_launch_algorithmic_write_GB']),
            'direct_equivalent_tflops': d.get('algorithmic_tflops'), 'avg_launch_ms': d.get('avg_launch_ms'),
            'avg_launch_gflop_algorithmic': d.get('avg_launch_gflop_algorithmic'),
            'timed_with': f'HIP events around every launch of {prof_steps} '
                          + ('eager step(s) run after the graph-replayed timed region' if args.graph else 'timed step(s)'),
            'all_conv_algorithmic_tflops': s['total_flops'] / (s['total_ms'] * 1e-3) / 1e12 if s['total_ms'] > 0 else None,
            'conv_time_share_of_step': s['total_ms'] * 1e-3 / prof_steps / step_s,
            'families': fam}


def run_lorenz_eval(args, device):
    """The reference's canonical workload, whole (experiments/lorenz/eval.py:72-84): six posterior-sampling runs of 1024 trajectories
    x 256 steps with C = 0, 1, 2, 4, 8, 16 Langevin corrections, tau = 0.25, event (65, 3), Gaussian guidance through
    A = x[..., ::step, :1] with std / step = 0.05 / 8 ("lo") or 0.25 / 1 ("hi"), gamma = 3e-2, for the global (lorenz/utils.py:26-42)
    or the local (:45-59) score network -- random-init inside the synthetic estimator (module docstring).  The reference budgets one
    such six-run job (plus its ground-truth particle filter) at <= 1 h on one GPU (eval.py:42).  `value` = all 6 x 256 diffusion
    steps / their wall-clock; every run is a full `VPSDE.sample` loop replayed from a captured hipGraph, initial draw on the host RNG
    as the reference (score.py:243), corrector noise from the device RNG."""
    from sda_amd import observe as Ob
    from sda_amd.experiments.lorenz import make_global_score, make_local_score
    from sda_amd.score import GaussianScore, VPSDE
    step, std = LORENZ_EVAL_FREQ[args.lorenz_freq]
    B, L, S, steps = args.per_gpu or 1024, 65, 3, 256
    torch.manual_seed(0)
    net = make_local_score() if args.lorenz_net == 'local' else make_global_score()
    score = SyntheticScore(net)
    inner = VPSDE(score, shape=())
    object.__setattr__(score, '_sched', inner)
    y = torch.randn((L + step - 1) // step, 1, generator=torch.Generator().manual_seed(2))      # one observation for the batch (eval.py:47)
    gs = GaussianScore(y, A=Ob.Subsample((slice(None, None, step), slice(0, 1))), std=std, sde=inner, gamma=3e-2)
    sde = VPSDE(gs, shape=(L, S)).to(device)
    # warm-up: weight packing, allocator pools, clocks
    wu = sde.sampler((B,), steps=steps, corrections=1, tau=0.25)
    for _ in range(max(args.warmup, 1)):
        wu.step()
    torch.cuda.synchronize(device)
    runs, loop_s, setup_s, finite, note = {}, 0.0, 0.0, True, None
    torch.manual_seed(1)
    for C in (0, 1, 2, 4, 8, 16):
        t0 = time.perf_counter()
        sampler = sde.sampler((B,), steps=steps, corrections=C, tau=0.25)
        if args.graph:
            try:
                sampler.capture()
            except Exception as e:  # noqa: BLE001
                note = f'capture failed, ran eagerly: {type(e).__name__}: {str(e)[:160]}'
                args.graph = 0
                sampler = sde.sampler((B,), steps=steps, corrections=C, tau=0.25)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(steps):
            sampler.step()
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        ok = bool(torch.isfinite(sampler.result()).all().item())
        finite = finite and ok
        runs[str(C)] = {'sample_s': t2 - t1, 'setup_s': t1 - t0, 'ms_per_step': (t2 - t1) / steps * 1e3,
                        'ms_per_score_eval': (t2 - t1) / steps / (1 + C) * 1e3, 'finite': ok}
        loop_s += t2 - t1
        setup_s += t1 - t0
    nsteps = 6 * steps
    out = {'metric': 'diffusion-steps/s (experiments/lorenz/eval.py protocol: six 256-step posterior-sampling runs, C = 0..16)',
           'value': nsteps / loop_s, 'unit': 'diffusion-steps/s', 'n_gpus': 1, 'steps': nsteps, 'warmup': max(args.warmup, 1),
           'ms_per_step': loop_s / nsteps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
           'data': 'synthetic (random-init net + exact Gaussian term, synthetic observation)',
           'config': {'workload': 'lorenz_eval', 'description': WORKLOADS['lorenz_eval']['desc'], 'event': [L, S], 'per_gpu_batch': B,
                      'global_batch': B, 'guided': True, 'corrections': [0, 1, 2, 4, 8, 16], 'tau': 0.25, 'schedule_steps': steps,
                      'gamma': 3e-2, 'net': args.lorenz_net, 'freq': args.lorenz_freq, 'obs_step': step, 'std': std,
                      'score_evals_total': steps * 37, 'hipgraph_step': bool(args.graph), 'hipgraph_note': note,
                      'observation': 'fused Subsample (hand-written adjoint; no autograd through A)',
                      'corrector_noise': 'device RNG (torch.randn_like under graph capture)', 'parallelism': 'dp1'},
           'six_run_wallclock_s': loop_s + setup_s, 'six_run_sampling_s': loop_s, 'six_run_setup_and_capture_s': setup_s,
           'reference_budget': 'one (network, freq) job of eval.py -- these six runs + its ground-truth particle filter -- is a Slurm '
                               'request of <= 1 h on one GPU (experiments/lorenz/eval.py:42); the reference publishes no timing',
           'per_C': runs, 'samples_finite': finite,
           'ms_per_score_eval_all_runs': loop_s / (steps * 37) * 1e3}
    if os.environ.get('SDA_HIP_LIB'):
        out['kernel_library_override'] = os.environ['SDA_HIP_LIB']
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = lorenz_eval_cpu(args, step, std, B, L, S, steps)
        out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
    print(json.dumps(out), flush=True)


def lorenz_eval_cpu(args, step, std, B, L, S, steps):
    """CPU leg of lorenz_eval: the oracle's loop (C = 1) on a bounded sample of rows, scaled by rows and by score evaluations (37
    evaluations per schedule step over the six runs; the cost of a step is its evaluations)."""
    from oracle import sda_oracle as O
    torch.manual_seed(0)
    sched = O.Schedule()
    rows = 64
    if args.lorenz_net == 'local':
        cfg = O.ResMLPConfig(15 + 32, 15, (128,) * 5, 'SiLU')
        sd = O.init_score_net(0, 'kernel.', cfg, 32)
        net = lambda xx, tt: O.mc_score_net(lambda a, b, c=None: O.score_net(sd, 'kernel.', cfg, a, b, c), 2, xx, tt)
    else:
        cfg = O.UNetConfig(S, S, 32, (64,), (3,), 3, 2, 'SiLU', 1, 'zeros')
        sd = O.init_score_unet(0, 'score.', cfg)
        net = lambda xx, tt: O.mc_score_wrapper(lambda a, b, c=None: O.score_unet(sd, 'score.', cfg, a, b, c), xx, tt)
    A = lambda v: v[..., ::step, :1]
    y = torch.randn((L + step - 1) // step, 1)

    def eps(xx, tt):
        mu, sg = sched.mu(tt), sched.sigma(tt)
        return xx * (sg / (mu * mu + sg * sg)) + 0.1 * net(xx, tt)
    score = lambda xx, tt: O.gaussian_score(eps, sched, y, A, std, 3e-2, xx, tt)
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, int(args.cpu_threads) or 32)
    torch.set_num_threads(cores)
    x = torch.randn(rows, L, S)
    O.sample(score, sched, x, 2, steps=2, corrections=1, tau=0.25)                      # warm-up
    nst, t_spent = 0, 0.0
    while nst < 3 or (t_spent < args.cpu_seconds and nst < 64):
        t0 = time.perf_counter()
        x = O.sample(score, sched, x, 2, steps=1, corrections=1, tau=0.25)
        t_spent += time.perf_counter() - t0
        nst += 1
    evals_per_s = 2 * nst / t_spent * rows / B                                           # guided evaluations of the full batch / s
    value = evals_per_s / (37 / 6)                                                       # 37 evaluations per 6 protocol steps
    return dict(value=value, unit='diffusion-steps/s (same protocol, extrapolated linearly in rows and score evaluations)', cores=cores,
                kind='port', cpu_model=_cpu_model(), host_logical_cpus=ncpu,
                sample=f'{nst} timed C = 1 steps after warm-up ({t_spent:.1f} s) of {rows} of the {B} trajectories, oracle loop '
                       f'(oracle/sda_oracle.py), {cores} threads; scaled by {rows}/{B} rows and 2 / (37/6) evaluations per step')


def partition(per_gpu, scaling, rank, world):
    """(rows of this rank, global batch, first global row).  weak: every rank owns one configuration shard (`per_gpu` rows; N = 8 is
    the BASELINE configuration itself).  strong: the configuration's global batch (per_gpu x 8) is fixed and split over the ranks."""
    from sda_amd import parallel
    if scaling == 'strong':
        global_batch = per_gpu * 8
        lo, hi = parallel.shard_range(global_batch, rank, world)
        return hi - lo, global_batch, lo
    return per_gpu, per_gpu * world, rank * per_gpu


def rank_inputs(wl, event, scaling, rank, world):
    """This rank's rows of the job's observation y and initial draw x(1), both keyed by GLOBAL row: any world size sees the same
    y and samples the same trajectories, ranks draw distinct noise, and no rank materialises rows it does not own."""
    from sda_amd import parallel
    b, global_batch, lo = partition(wl['per_gpu'], scaling, rank, world)
    oshape = (event[0], event[1], event[2] // 4, event[3] // 4) if wl['kind'] == 'kolmogorov' else ((event[0] + 7) // 8, 1)
    if scaling == 'strong':
        return (parallel.sharded_initial_noise(global_batch, oshape, 2, rank, world),
                parallel.sharded_initial_noise(global_batch, event, 1, rank, world))
    y = torch.stack([torch.randn(oshape, generator=torch.Generator().manual_seed(2000003 + lo + i)) for i in range(b)])
    gen = torch.Generator()
    rows = []
    for i in range(lo, lo + b):
        gen.manual_seed((1 * 1000003 + i) & 0x7fffffffffffffff)
        rows.append(torch.randn(tuple(event), generator=gen))
    return y, torch.stack(rows)


def launch_plan(gpus, env, argv):
    """argv of the torch.distributed.run launcher this process re-executes itself under, or None when it already IS a rank
    (WORLD_SIZE set: launched by torchrun / the driver's `python -m torch.distributed.run ...` form) or a 1-GPU job.
    The reference has no multi-process launcher (experiments/lorenz/eval.py:42: Slurm job arrays); one process per GPU over
    RCCL is this project's data-parallel form."""
    if gpus <= 1 or env.get('WORLD_SIZE') is not None:
        return None
    port = env.get('MASTER_PORT')
    if port is None:
        import socket
        with socket.socket() as so:                  # a free port: concurrent jobs on one node must not collide
            so.bind(('127.0.0.1', 0))
            port = str(so.getsockname()[1])
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def second_line(args, sde, sampler, b, device, f32_s_per_step):
    """The OPT-IN line: the same captured step with the block convolutions on the f16 matrix cores (csrc/conv_h2.hip: every fp32
    operand as two halves, three products, fp32 accumulation), timed like the headline (graph replay, sync on both sides), plus one
    eager step from the SAME state and noise under both multiplies -- how far apart the two arithmetic routes land."""
    from sda_amd import ops
    kw = dict(steps=1000, corrections=args.corrections, tau=args.tau)
    try:
        sampler._graph = None
        del sampler
        torch.cuda.empty_cache()
        ops.set_multiply('f32')
        sa = sde.sampler((b,), **kw)
        sa.step()
        xa = sa.x.clone()
        del sa
        ops.set_multiply('f16x2')
        sb = sde.sampler((b,), **kw)
        sb.step()
        diff = ((sb.x - xa).abs().max() / xa.abs().max()).item()
        del sb, xa
        torch.cuda.empty_cache()
        s2 = sde.sampler((b,), **kw)
        note = None
        if args.graph:
            try:
                s2.capture()
            except Exception as e:  # noqa: BLE001
                note = f'capture failed, ran eagerly: {type(e).__name__}: {str(e)[:160]}'
                s2 = sde.sampler((b,), **kw)
        for _ in range(max(1, args.second_warmup)):
            s2.step()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(args.second_steps):
            s2.step()
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / args.second_steps
        return {'multiply': 'f16x2', 'dtype': 'f32 emulated as 2 x f16 (three f16 products per multiply on v_mfma_f32_32x32x16_f16, fp32 accumulate)',
                'ms_per_step': dt * 1e3, 'value': 1.0 / dt, 'unit': 'diffusion-steps/s', 'steps': args.second_steps, 'warmup': max(1, args.second_warmup),
                'speedup_vs_f32_headline': f32_s_per_step / dt, 'samples_finite': bool(torch.isfinite(s2.x).all().item()),
                'one_step_max_abs_diff_vs_f32_over_max_abs': diff, 'hipgraph_note': note,
                'note': 'OPT-IN (ops.set_multiply / SDA_MULTIPLY=f16x2); `value` above is the fp32-MFMA headline'}
    except Exception as e:  # noqa: BLE001 -- the second line must never cost the headline
        return {'multiply': 'f16x2', 'error': f'{type(e).__name__}: {str(e)[:300]}'}
    finally:
        ops.set_multiply('f32')


OTHER_CONFIGS = (('lorenz63', 'configs[0]', 20, 100), ('lorenz96', 'configs[1]', 20, 100), ('kolmogorov64', 'configs[2]', 1, 5),
                 ('qg128', 'configs[4]', 1, 5), ('kolmogorov64_default', "the reference's default widths (64, 128, 256)", 1, 5))


def other_configs(args, device):
    """Every OTHER BASELINE configuration, bounded, in the same JSON line (`other_configs`): the same job as `--workload NAME` at N = 1
    (same model, inputs, guidance, corrections, captured hipGraph), `warmup` untimed + `steps` timed replays bracketed by device syncs,
    then one eager step under HIP events for the dominant kernel family and its fraction of its roofline.  Runs after the headline's
    timed region and never touches it; a failure is reported in place of the entry."""
    from sda_amd import ops, parallel
    from sda_amd.score import GaussianScore, VPSDE
    out, t_all = {}, time.perf_counter()
    for name, label, warm, steps in OTHER_CONFIGS:
        t_cfg = time.perf_counter()
        try:
            wl = dict(WORKLOADS[name])
            b, _, lo = partition(wl['per_gpu'], 'weak', 0, 1)
            net, event, A = build_model(wl, device)
            score = SyntheticScore(net)
            inner = VPSDE(score, shape=())
            object.__setattr__(score, '_sched', inner)
            y, x_init = rank_inputs(wl, event, 'weak', 0, 1)
            eps_mod = GaussianScore(y, A=A, std=0.1, sde=inner) if args.guided else score
            sde = VPSDE(eps_mod, shape=event).to(device)
            sde.initial_noise = x_init
            if args.corrections > 0:
                sde.noise_source = parallel.KeyedNoise((lo, lo + b), event, 2, args.corrections, device)
            sampler = sde.sampler((b,), steps=1000, corrections=args.corrections, tau=args.tau)
            graph = True
            try:
                sampler.capture()
            except Exception:  # noqa: BLE001
                graph = False
                sampler = sde.sampler((b,), steps=1000, corrections=args.corrections, tau=args.tau)
            for _ in range(warm):
                sampler.step()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                sampler.step()
            torch.cuda.synchronize(device)
            dt = (time.perf_counter() - t0) / steps
            sampler._graph = None
            prof = ops.ConvProfile()
            ops.conv_profile = prof
            sampler.step()
            torch.cuda.synchronize(device)
            ops.conv_profile = None
            a2 = argparse.Namespace(**vars(args))
            a2.workload = name
            rf = roofline_report(prof, 1, dt, a2, ROOT, None)
            fams = rf['families']
            conv = {k: v for k, v in fams.items() if 'issued_mfma_tflops' in v}
            dom = max(conv, key=lambda k: conv[k]['ms_per_step']) if conv else None
            entry = {'config': label, 'description': wl['desc'], 'ms_per_step': dt * 1e3, 'value': 1.0 / dt, 'unit': 'diffusion-steps/s',
                     'steps': steps, 'warmup': warm, 'hipgraph_step': graph, 'per_gpu_batch': b, 'guided': bool(args.guided),
                     'corrections': args.corrections, 'samples_finite': bool(torch.isfinite(sampler.x).all().item()),
                     'dominant_family': dom, 'dominant_kernel': rf['kernel'], 'bound': rf['bound'],
                     'dominant_share_of_step': None if dom is None else conv[dom]['share_of_step'],
                     'dominant_mfma_frac_issued': None if dom is None else conv[dom]['mfma_util'],
                     'dominant_algorithmic_tflops': None if dom is None else conv[dom]['algorithmic_tflops'],
                     'all_conv_algorithmic_tflops': rf['all_conv_algorithmic_tflops'],
                     'families': {k: {'share_of_step': v['share_of_step'], 'ms_per_step': v['ms_per_step'],
                                      **({'algorithmic_tflops': v['algorithmic_tflops'], 'mfma_util': v['mfma_util']} if 'mfma_util' in v else
                                         {'frac_of_8TBps': v.get('frac_of_8TBps')})} for k, v in fams.items()},
                     'wall_s': None}
            del sampler, sde, eps_mod, inner, score, net, prof
        except Exception as e:  # noqa: BLE001 -- the extra lines must never cost the headline
            ops.conv_profile = None
            entry = {'config': label, 'error': f'{type(e).__name__}: {str(e)[:300]}'}
        torch.cuda.empty_cache()
        entry['wall_s'] = time.perf_counter() - t_cfg
        out[name] = entry
    out['total_wall_s'] = time.perf_counter() - t_all
    out['note'] = ('same job as `bench.py --workload NAME` at N = 1, fewer timed steps; fp32 multiply; timed outside the headline region; '
                   'dominant_mfma_frac_issued = issued fp32 MFMA flops / time / 157.3 TFLOP/s (latency-bound 1-D nets: see `bound`)')
    return out


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
    ap.add_argument('--cpu-seconds', type=float, default=20.0, help='timed CPU steps continue past the minimum of 3 while under this budget')
    ap.add_argument('--cpu-windows', type=int, default=1, help='trajectory windows of the CPU sample (Kolmogorov workloads)')
    ap.add_argument('--cpu-threads', type=int, default=0, help='0 = calibrate: fastest of 16..128 threads on one score evaluation of the sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--graph', type=int, default=1, help='1 (default): replay each step from a captured hipGraph')
    ap.add_argument('--profile-steps', type=int, default=1, help='extra eager steps (outside the timed region) for the per-kernel HIP-event timings when --graph 1')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'], help="weak: every rank owns a configuration shard (N = 8 is the configuration itself); strong: the 8-shard global batch is split over the ranks")
    ap.add_argument('--lorenz-net', default='global', choices=['global', 'local'], help='lorenz_eval: the score network (lorenz/utils.py:26-59)')
    ap.add_argument('--lorenz-freq', default='lo', choices=['lo', 'hi'], help='lorenz_eval: observation setting (eval.py:50-53)')
    ap.add_argument('--multiply', default='f32', choices=['f32', 'f16x2'],
                    help="f32 (default, the headline): fp32 MFMAs.  f16x2 (OPT-IN second line): the block convolutions multiply on the f16 matrix "
                         "cores, every fp32 operand as two halves, three products, fp32 accumulation (csrc/conv_h2.hip)")
    ap.add_argument('--second-line', type=int, default=1,
                    help="1 (default; N = 1, Kolmogorov-shaped workloads, --multiply f32): after the f32 headline, time the same step with "
                         "--multiply f16x2 and report it as the `opt_in_f16x2` object of the same JSON line (never `value`)")
    ap.add_argument('--second-steps', type=int, default=8)
    ap.add_argument('--second-warmup', type=int, default=1)
    ap.add_argument('--other-configs', type=int, default=-1,
                    help="1: add the bounded `other_configs` object (every other BASELINE configuration + the reference's default-width net, "
                         "<= ~60 s in total) to the line; default: on for the default workload at N = 1, off otherwise")
    ap.add_argument('--force-pg', type=int, default=0,
                    help="1: initialise the process group (and issue the final all-gather through it) even at WORLD_SIZE = 1 -- first contact "
                         "of RCCL, the HSA IPC setting and libsda_hip.so in one process on a 1-GPU box (tests/test_gpu_rccl.py)")
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' only to exercise the launch path on a 1-GPU box)")
    args = ap.parse_args()

    # (the host driver of this pool only supports dmabuf IPC: without this RCCL's hipIpcGetMemHandle fails; normally exported already)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    plan = launch_plan(args.gpus, os.environ, sys.argv[1:])
    if plan is not None:
        # `python bench.py --gpus N` (the driver's verbatim call, no torchrun): this process becomes the launcher of N ranks
        if args.backend == 'nccl' and torch.cuda.device_count() < args.gpus:
            sys.exit(f'bench.py: --gpus {args.gpus} with backend nccl (RCCL) needs {args.gpus} visible GPUs, found '
                     f'{torch.cuda.device_count()} (one rank per GPU; `--backend gloo` only exercises the launch path on a smaller box)')
        os.execv(sys.executable, plan)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)            # one rank per GPU on a real node; wraps only in 1-GPU dry runs
    device = torch.device('cuda', dev_index)
    torch.cuda.set_device(device)
    pg_live = world > 1 or bool(args.force_pg)
    if pg_live:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)                          # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    if args.gpus != world:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks (pass the same N to both)')

    from sda_amd import ops, parallel
    from sda_amd.score import GaussianScore, VPSDE
    ops.set_multiply(args.multiply)

    if args.workload == 'lorenz_eval':
        if world != 1:
            sys.exit('bench.py: --workload lorenz_eval is a single-GPU protocol (the reference runs it as independent Slurm array jobs)')
        return run_lorenz_eval(args, device)
    wl = dict(WORKLOADS[args.workload])
    if args.per_gpu:
        wl['per_gpu'] = args.per_gpu
    b, global_batch, lo = partition(wl['per_gpu'], args.scaling, rank, world)
    net, event, A = build_model(wl, device)
    score = SyntheticScore(net)
    inner = VPSDE(score, shape=())
    object.__setattr__(score, '_sched', inner)       # plain attribute: not a submodule (inner.eps is score)
    y, x_init = rank_inputs(wl, event, args.scaling, rank, world)
    eps_mod = GaussianScore(y, A=A, std=0.1, sde=inner) if args.guided else score
    sde = VPSDE(eps_mod, shape=event).to(device)
    sde.initial_noise = x_init
    if args.corrections > 0:
        sde.noise_source = parallel.KeyedNoise((lo, lo + b), event, 2, args.corrections, device)
    sampler = sde.sampler((b,), steps=1000, corrections=args.corrections, tau=args.tau)

    def sync():
        torch.cuda.synchronize(device)
        if pg_live:
            dist.barrier()
            torch.cuda.synchronize(device)           # the RCCL barrier is itself stream work

    graph_note = None
    if args.graph:
        try:
            sampler.capture()
        except Exception as e:  # noqa: BLE001 -- e.g. the graph's private pool does not fit next to the eager allocations
            graph_note = f'capture failed, ran eagerly: {type(e).__name__}: {str(e)[:200]}'
            args.graph = 0
            torch.cuda.synchronize(device)
            sampler = sde.sampler((b,), steps=1000, corrections=args.corrections, tau=args.tau)
    for _ in range(args.warmup):
        sampler.step()
    sync()
    probe = None
    if rank == 0 and not args.no_profile:
        probe = ops.clock_probe(device)              # (before the timed region; synchronises)
    if world > 1:
        sync()
    prof = None
    if rank == 0 and not args.no_profile and not args.graph:
        prof = ops.ConvProfile()                     # eager timed region: the events bracket the timed launches themselves
        ops.conv_profile = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sampler.step()
    sync()
    elapsed = time.perf_counter() - t0
    ops.conv_profile = None
    if pg_live:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    prof_steps = 0
    if rank == 0 and not args.no_profile and args.graph and args.profile_steps > 0:
        # events cannot bracket launches inside a graph replay: time the same kernels in extra EAGER steps, outside `value`
        sampler._graph = None
        prof = ops.ConvProfile()
        ops.conv_profile = prof
        for _ in range(args.profile_steps):
            sampler.step()
        torch.cuda.synchronize(device)
        ops.conv_profile = None
        prof_steps = args.profile_steps
    elif prof is not None:
        prof_steps = args.steps

    finite = bool(torch.isfinite(sampler.x).all().item())
    # the one collective of the job: gather the samples (after the loop; not part of a step)
    t1 = time.perf_counter()
    gathered = parallel.all_gather_samples(sampler.x, global_batch, always_collective=pg_live)
    sync()
    gather_ms = (time.perf_counter() - t1) * 1e3
    assert gathered.shape[0] == global_batch

    if rank == 0:
        # unit of `value`: one diffusion step of one configuration shard (per_gpu trajectories); weak scaling: every rank
        # advances one shard per step; strong scaling: the job advances the 8-shard global batch once per step
        shards_per_step = world if args.scaling == 'weak' else 8
        value = shards_per_step * args.steps / elapsed
        out = {
            'metric': 'diffusion-steps/s (Kolmogorov 64x256x256 posterior sampling; unit = one predictor-corrector step of a '
                      '16-trajectory per-GPU shard)' if args.workload == 'kolmogorov256' else f'diffusion-steps/s ({args.workload})',
            'value': value, 'unit': 'diffusion-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32' if args.multiply == 'f32' else 'f32 emulated as 2 x f16 (three f16 products per multiply on v_mfma_f32_32x32x16_f16, fp32 accumulate; OPT-IN, not the headline)',
            'multiply': args.multiply,
            'data': 'synthetic (random-init net + exact Gaussian term, synthetic observations)',
            'config': {'workload': args.workload, 'description': wl['desc'], 'event': list(event),
                       'per_gpu_batch': b, 'global_batch': global_batch, 'guided': bool(args.guided),
                       'corrections': args.corrections, 'tau': args.tau, 'schedule_steps': 1000,
                       'score_evals_per_step': 1 + args.corrections, 'hipgraph_step': bool(args.graph), 'hipgraph_note': graph_note,
                       'observation': 'fused Subsample (hand-written adjoint; no autograd through A)',
                       'corrector_noise': 'row-keyed Philox (sda_randn_rows)',
                       'parallelism': f'dp{world} (batch-sharded, no in-loop collective)',
                       'ranks_seen': dist.get_world_size() if pg_live else 1, 'process_group_live': pg_live,
                       'backend': (args.backend + (' (RCCL)' if args.backend == 'nccl' else '')) if pg_live else None},
            'wallclock_per_1000_steps_s': elapsed / args.steps * 1000,
            'samples_finite': finite, 'final_allgather_ms': gather_ms,
        }
        if os.environ.get('SDA_HIP_LIB'):            # a tooling build of the kernels was swapped in: say so in the line
            out['kernel_library_override'] = os.environ['SDA_HIP_LIB']
        if prof is not None and prof_steps > 0:
            out['roofline'] = roofline_report(prof, prof_steps, elapsed / args.steps, args, ROOT, probe)
        if world == 1 and args.second_line and args.multiply == 'f32' and wl['kind'] == 'kolmogorov':
            out['opt_in_f16x2'] = second_line(args, sde, sampler, b, device, elapsed / args.steps)
        want_other = args.other_configs == 1 or (args.other_configs < 0 and args.workload == 'kolmogorov256' and args.multiply == 'f32')
        if world == 1 and want_other:
            sampler = gathered = None                # (the shard's state, its graph pool and the gathered copy: 50+ GB back to the allocator)
            sde.initial_noise = None
            torch.cuda.empty_cache()
            out['other_configs'] = other_configs(args, device)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(wl, args, bool(args.guided), args.corrections)
            out['gpu_over_cpu'] = value / out['cpu_baseline']['value']
        print(json.dumps(out), flush=True)
    if pg_live:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
