r"""Evaluation metrics of the sampling experiments: the reference's ``sda.utils.emd`` / ``mmd`` / ``bpf``
(sda/utils.py:168-263), SURVEY section 8(f)-4.  Same names, arguments and return values.

* ``emd`` -- the pairwise Euclidean cost matrix is a HIP kernel (``sda_pairwise_dist``); the transport problem itself is
  solved on the host, as in the reference (``ot.emd2`` is POT's CPU network simplex; POT is a third-party dependency the
  reference does not vendor).  With empty weight vectors POT uses uniform weights, and for equally many samples on both
  sides -- how experiments/lorenz/eval.py:61-63,89 call it (1024 vs 1024) -- an optimal plan is a permutation, so the LP
  is a linear assignment problem (``sda_assignment_cost``, exact, O(n^3)).  Unequal sample counts go through an integral
  min-cost flow on the host (``sda_transport_cost``), exact as well.
* ``mmd`` -- squared distances from the same kernel (differences are squared directly instead of expanding
  |x|^2 + |y|^2 - 2 x.y, which removes the reference's fp32 cancellation noise on the small bandwidths), then one fused
  pass per Gram block accumulates the seven Gaussian kernels in float64.
* ``bpf`` -- bootstrap particle filter around user callables (``transition``, ``likelihood``); the trajectory buffer is
  allocated once and ancestors are resampled by index instead of re-concatenating the history at every step.
"""
from typing import Callable

import torch
from torch import Tensor

from . import ops
from ._lib import SdaHipError


def emd(x: Tensor, y: Tensor) -> Tensor:
    r"""Earth mover's distance between two equally weighted sample sets ``x`` (M, \*) and ``y`` (N, \*)
    (sda/utils.py:203-219: ``ot.emd2`` with empty weight vectors = uniform marginals).  The cost matrix is formed on the
    device; the transport LP is solved on the host (as POT does): a linear assignment for M == N, an integral min-cost flow
    otherwise.

    Size note: the M == N solve is O(N^3) shortest augmenting paths (1024 vs 1024 trajectories: ~0.1 s).  The M != N solve is
    successive shortest paths with DENSE Dijkstra over the complete bipartite graph -- O((M + N)^2) per augmentation, at
    least M + N augmentations, and an M x N int64 flow matrix: fine for the sample counts of the reference's evaluation
    (hundreds to a couple of thousand), minutes beyond ~2000 x 1000 on one host thread (POT's network simplex takes
    seconds there).  Past that size prefer equal sample counts."""
    xf, yf = x.flatten(1).float(), y.flatten(1).float()
    cost = ops.pairwise_dist(xf, yf, squared=False).cpu()
    if xf.shape[0] == yf.shape[0]:
        total, _ = ops.assignment_cost(cost)
        return x.new_tensor(total / xf.shape[0])
    return x.new_tensor(ops.transport_cost(cost))


def mmd(x: Tensor, y: Tensor) -> Tensor:
    r"""Empirical maximum mean discrepancy with Gaussian kernels of bandwidth 1e-3 ... 1e3 (sda/utils.py:222-263)."""
    xf, yf = x.flatten(1).float(), y.flatten(1).float()
    m, n = xf.shape[0], yf.shape[0]
    kxx = ops.mmd_kernel_sum(ops.pairwise_dist(xf, xf, squared=True)) / (m * m)
    kyy = ops.mmd_kernel_sum(ops.pairwise_dist(yf, yf, squared=True)) / (n * n)
    kxy = ops.mmd_kernel_sum(ops.pairwise_dist(xf, yf, squared=True)) / (m * n)
    return (kxx + kyy - 2 * kxy).to(x.dtype)


def bpf(x: Tensor, y: Tensor, transition: Callable[[Tensor], Tensor], likelihood: Callable[[Tensor, Tensor], Tensor],
        step: int = 1) -> Tensor:
    r"""Bootstrap particle filter (sda/utils.py:168-200): ``x`` (M, \*) initial particles, ``y`` (N, \*) observations;
    returns the (M, N*step + 1, \*) resampled trajectories.  ``likelihood(y_i, x_i)`` returns normalised weights (M,)."""
    m, n = x.shape[0], len(y)
    traj = torch.empty((m, n * step + 1) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    traj[:, 0] = x
    filled = 1
    for yi in y:
        for _ in range(step):
            traj[:, filled] = transition(traj[:, filled - 1])
            filled += 1
        w = likelihood(yi, traj[:, filled - 1])
        j = torch.multinomial(w, len(w), replacement=True)
        traj[:, :filled] = traj[j, :filled]          # whole histories follow their resampled ancestor
    return traj
