#!/usr/bin/env python3 -B
"""Fixture for the LONG free-running parity chain at the reference network size: the oracle's own sampling loop
(oracle/sda_oracle.py: sample + gaussian_score, restating sda/score.py:225-263, :375-396) on the reference Kolmogorov net
(experiments/kolmogorov/train.py:15-22: window 5, (96, 192, 384), (3, 3, 3), 22 874 922 parameters) at 64 x 64, trajectories of L = 6,
Gaussian guidance through x[..., ::4, ::4] (std 0.1, gamma 1e-2), 32 diffusion steps with one Langevin correction each (tau 0.5) = 64
guided evaluations deep, in fp32 AND fp64, from seeded inputs.  One guided evaluation costs the host 2.5 s (fp32) / 5 s (fp64) per
trajectory, so the routine GPU suite could only afford 8 steps; with this fixture the GPU box runs the HIP side alone and the 32-step
form is part of every `-m gpu` run (tests/test_gpu_kolmogorov_eval.py).

Everything is regenerated from seeds on both sides (net: torch.manual_seed(70) + make_score; x1 / y: manual_seed(71 + batch);
corrector noise: a torch CPU generator, seed 7700 + batch) -- the fixture carries only the oracle's final samples and float64
digests of every input, which the test checks before it trusts the references (a torch version whose CPU generator or default
initialisers differ fails loudly as 'stale fixture', not as a parity error).  Data only; no reference source involved -- the oracle
is pinned to the reference by tests/test_oracle_golden.py.

    python3 -B tests/golden/make_golden_k64_chain.py [--batches 2,1] [--steps 32] [--threads 8]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

K64 = dict(window=5, embedding=64, hidden_channels=(96, 192, 384), hidden_blocks=(3, 3, 3))
EVENT = (6, 2, 64, 64)
CORR, TAU, STD, GAMMA = 1, 0.5, 0.1, 1e-2


def digest(t: torch.Tensor):
    d = t.detach().double().reshape(-1)
    w = torch.arange(1, d.numel() + 1, dtype=torch.float64) % 8191 + 1          # position-weighted: a permutation changes it
    return np.array([d.sum().item(), d.abs().sum().item(), (d * w).sum().item()])


def chain_inputs(batch: int, steps: int):
    """(x1, y, zs): the test rebuilds exactly these (tests/test_gpu_kolmogorov_eval.py::_chain_inputs)."""
    torch.manual_seed(71 + batch)
    x1 = torch.randn((batch,) + EVENT)
    y = torch.randn(x1[0][..., ::4, ::4].shape)
    g = torch.Generator().manual_seed(7700 + batch)
    zs = torch.randn((steps * CORR, batch) + EVENT, generator=g)
    return x1, y, zs


def k64_net():
    from sda_amd.experiments.kolmogorov import make_score
    torch.manual_seed(70)
    net = make_score(size=64, **K64)
    assert sum(p.numel() for p in net.parameters()) == 22_874_922
    return net


def stats_fixture(B: int):
    """The oracle side of tests/test_gpu_kolmogorov_eval.py::test_kolmogorov_assimilation_statistical_end_to_end at B trajectories
    (the reference's own acceptance check `(A(x) - y).std()`, figures.ipynb#cell11, as per-trajectory log-spreads of two independent
    runs of the oracle's loop on the N(0, I) prior): only the statistics the test compares are kept, not the samples."""
    import math
    from oracle import sda_oracle as O
    L, steps, corr, tau, std, gamma = 6, 128, 1, 0.5, 0.1, 1e-2
    event = (L, 2, 64, 64)
    sub4 = lambda v: v[..., ::4, ::4]
    torch.manual_seed(80)
    y = torch.randn(sub4(torch.empty(event)).shape) * math.sqrt(1 + std ** 2)
    sched = O.Schedule()
    eta = 1e-3
    eps = lambda x, t: x * (sched.sigma(t) / (1 + eta * eta))
    score = lambda x, t: O.gaussian_score(eps, sched, y, sub4, std, gamma, x, t)
    mask = torch.ones(64, 64, dtype=torch.bool)
    mask[::4, ::4] = False
    out = {'y_digest': digest(y), 'B': np.array(B), 'torch_version': np.array(torch.__version__)}
    for tag, seed in (('a', 81), ('b', 82)):
        t0 = time.time()
        torch.manual_seed(seed)
        ref = O.sample(score, sched, torch.randn((B,) + event), 4, steps, corr, tau)
        obs = sub4(ref)
        out['log_spread_' + tag] = (obs - y).flatten(1).std(dim=1).log().numpy()
        out['spread_' + tag] = np.array((obs - y).std().item())
        out['unobserved_var_' + tag] = np.array(ref[..., mask].var().item())
        print(f'stats run {tag} (B = {B}): {time.time() - t0:.0f} s', flush=True)
    np.savez(os.path.join(HERE, f'k64_stats_b{B}.npz'), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stats', type=int, default=0, help='B: write k64_stats_b<B>.npz (the statistical test\'s oracle side) instead')
    ap.add_argument('--batches', default='2,1')
    ap.add_argument('--steps', type=int, default=32)
    ap.add_argument('--threads', type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    if args.stats:
        return stats_fixture(args.stats)
    from oracle import sda_oracle as O
    from tests.util import oracle_eps_from_module
    net = k64_net()
    eps_net = oracle_eps_from_module(net, 'mc2d')
    pdig = digest(torch.cat([p.detach().reshape(-1) for p in net.parameters()]))
    sched = O.Schedule()
    sub4 = lambda v: v[..., ::4, ::4]
    for batch in [int(b) for b in args.batches.split(',')]:
        x1, y, zs = chain_inputs(batch, args.steps)
        out = {'params_digest': pdig, 'x1_digest': digest(x1), 'y_digest': digest(y), 'zs_digest': digest(zs),
               'steps': np.array(args.steps), 'batch': np.array(batch), 'torch_version': np.array(torch.__version__)}
        for name, dtype in (('ref32', torch.float32), ('ref64', torch.float64)):
            def eps(xx, tt):
                mu, sg = sched.mu(tt), sched.sigma(tt)
                return xx * (sg / (mu * mu + sg * sg)) + 0.1 * eps_net(xx, tt, None if dtype == torch.float32 else dtype)
            sc = lambda xx, tt: O.gaussian_score(eps, sched, y.to(dtype), sub4, STD, GAMMA, xx, tt)
            zz = zs.to(dtype)
            t0 = time.time()
            rec = []
            ref = O.sample(sc, sched, x1.to(dtype), 4, args.steps, CORR, TAU, noise=lambda i, j: zz[i * CORR + j], record=rec)
            print(f'batch {batch} {name}: {time.time() - t0:.0f} s, finite {bool(torch.isfinite(ref).all())}', flush=True)
            out[name] = ref.numpy()
            if name == "ref32":
                out[name + "_step8"] = rec[7].numpy()              # the fp32 state after 8 steps as well (the former routine form)
        a, b = torch.from_numpy(out['ref32']).double(), torch.from_numpy(out['ref64'])
        own = ((a - b).abs().max() / b.abs().max()).item()
        out['own_fp32_vs_fp64'] = np.array(own)
        print(f'batch {batch}: fp32 oracle vs fp64 oracle after {args.steps} steps {own:.2e}', flush=True)
        np.savez(os.path.join(HERE, f'k64_chain_b{batch}.npz'), **out)


if __name__ == '__main__':
    main()
