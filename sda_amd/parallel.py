"""Multi-GPU posterior sampling: independent trajectories shard across ranks (one process per GPU,
``torch.distributed`` backend "nccl" = RCCL over xGMI), no communication inside the diffusion loop, one all-gather
of the final samples.

The reference has no distributed code at all (its only parallelism is Slurm job arrays,
experiments/lorenz/eval.py:42); this module is what a data-parallel launch of ``VPSDE.sample`` needs:
  * a deterministic batch partition,
  * per-rank noise that is a slice of the single-process noise stream, so 1-GPU and N-GPU runs draw the same noise, bit for
    bit, and hence the same trajectories up to fp32 round-off.  (The samples themselves are bit-identical across world sizes
    only while every shard takes the same kernel variants: tile sizes of the 1-D kernels, the LayerNorm kernel flavour and the
    conv_small1d / staged-kernel choice depend on the LOCAL batch, and each variant sums in its own order.  The tests' shapes
    stay on one side of every threshold; in general expect agreement to ~1e-6 relative, not equality.)
  * the final gather.
``DPSGaussianScore`` couples the batch through one scalar (``err`` summed over every sample, score.py:339-342): the one real
exchange step on the path.  ``sample_sharded`` gives such a score its shard and process group, and each evaluation all-reduces
that scalar (4 bytes per rank; RCCL on the GPU) -- a sharded DPS run then equals the single-process run like every other score.
Every rank must hold at least one trajectory then (``batch >= world_size``), so that all ranks enter every all-reduce.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of ``range(total)``: the first ``total % world`` ranks get one extra."""
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_initial_noise(batch: int, event: tuple, seed: int, rank: int, world_size: int) -> Tensor:
    """This rank's rows of the (batch, *event) initial draw (host RNG, as score.py:243).  Row i comes from its own
    generator stream keyed by (seed, i), so the union over ranks is the same tensor for every world size -- and no rank
    ever materialises rows it does not own (config [3]: 4.3 GB of noise for the global batch)."""
    lo, hi = shard_range(batch, rank, world_size)
    gen = torch.Generator()
    rows = []
    for i in range(lo, hi):
        gen.manual_seed((seed * 1000003 + i) & 0x7fffffffffffffff)
        rows.append(torch.randn(tuple(event), generator=gen))
    if not rows:
        return torch.empty((0,) + tuple(event))
    return torch.stack(rows)


class KeyedNoise:
    """Corrector noise ``z`` (score.py:257) for this rank's rows, keyed per trajectory: row ``lo + r`` of draw
    ``step * corrections + correction`` is a function of (seed, global row, draw) only (``sda_randn_rows``: Philox4x32-10 +
    Box-Muller on the device).  The union over ranks is therefore independent of the world size, every rank generates
    exactly its own rows -- nothing scales with the global batch -- and, since the draw index can be read from the
    sampler's device-side step counter, a sharded diffusion step stays hipGraph-capturable (``graph_safe``)."""

    graph_safe = True

    def __init__(self, rows: Tuple[int, int], event: tuple, seed: int, corrections: int, device):
        self.lo, self.hi = rows
        self.event, self.seed, self.corrections = tuple(event), int(seed), max(int(corrections), 1)
        self.device = device

    def _out(self) -> Tensor:
        return torch.empty((self.hi - self.lo,) + self.event, device=self.device, dtype=torch.float32)

    def __call__(self, step: int, correction: int) -> Tensor:
        from . import ops
        return ops.randn_rows(self._out(), self.seed, self.lo, draw=step * self.corrections + correction)

    def draw_dev(self, step_dev: Tensor, correction: int) -> Tensor:
        """Same draw with the step index read from device memory (graph replay)."""
        from . import ops
        return ops.randn_rows(self._out(), self.seed, self.lo, draw_dev=step_dev, draw_mul=self.corrections,
                              draw_add=correction)


class TableNoise:
    """Corrector noise replayed from a recorded table ``(steps * corrections, batch, *event)`` on the device: the injected-noise form of
    the reference's parity protocol (SURVEY 8c tiers 2 / 3: both sides of a free-running comparison consume the SAME draws), made
    graph-safe -- the row is selected by the device step counter, so a captured step replays it (``KeyedNoise`` generates its draws in
    the kernel; this source plays back draws made elsewhere, e.g. by the host generator a committed oracle fixture was made with)."""

    graph_safe = True

    def __init__(self, table: Tensor, corrections: int):
        self.table, self.corrections = table, max(int(corrections), 1)

    def __call__(self, step: int, correction: int) -> Tensor:
        return self.table[step * self.corrections + correction]

    def draw_dev(self, step_dev: Tensor, correction: int) -> Tensor:
        return self.table.index_select(0, step_dev.reshape(1) * self.corrections + correction)[0]


def all_gather_samples(local: Tensor, batch: int, always_collective: bool = False) -> Tensor:
    """Concatenate every rank's samples along dim 0 (ranks may hold unequal shares).  One collective, after the loop.

    ``always_collective``: issue the collective even in a one-rank group (a copy otherwise skipped) -- how a 1-GPU box exercises
    the RCCL call the 8-GPU job ends with (tests/test_gpu_rccl.py, ``bench.py --force-pg 1``).

    Equal shares (every configuration of BASELINE.json): one ``all_gather_into_tensor`` straight into the result -- no padding
    copy, no list of per-rank buffers (537 MB per rank at configs[3]).  Unequal shares: padded list gather."""
    rank, ws = world()
    if ws == 1 and not (always_collective and dist.is_available() and dist.is_initialized()):
        return local
    sizes = [shard_range(batch, r, ws) for r in range(ws)]
    counts = [hi - lo for lo, hi in sizes]
    if min(counts) == max(counts):
        out = torch.empty((batch,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    biggest = max(counts)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


_UNSET = object()


def sample_sharded(sde, batch: int, c: Optional[Tensor] = None, steps: int = 64, corrections: int = 0, tau: float = 1.0,
                   seed: int = 0, gather: bool = True, rank: Optional[int] = None,
                   world_size: Optional[int] = None) -> Tensor:
    """``sde.sample((batch,), ...)`` with the batch split over the ranks of the default process group: this rank's rows of
    the row-keyed noise streams, no collective inside the loop, one all-gather of the samples at the end.

    ``rank`` / ``world_size`` override the process group (and imply ``gather=False``): a single process can then compute
    any rank's shard, which is how the one-GPU test checks that the shards of a 2-rank job concatenate to the 1-rank job
    (bit for bit at the tests' sizes; up to fp32 round-off when shard and whole batch select different kernel variants, see
    the module docstring)."""
    emulated = not (rank is None or world_size is None)
    if emulated:
        gather = False
    else:
        rank, world_size = world()
    lo, hi = shard_range(batch, rank, world_size)
    event = tuple(sde.shape)
    coupled = _batch_coupled(sde)
    if coupled and emulated and world_size > 1:
        raise ValueError('DPSGaussianScore sums its error over the whole batch (score.py:339): an emulated rank has no peers '
                         'to all-reduce with; run it under a process group')
    if coupled and batch < world_size:
        raise ValueError(f'a batch-coupled score needs every rank in every all-reduce: batch {batch} < world size {world_size}')
    # everything this function touches on the caller's objects is snapshotted first and restored whatever happens
    keep = (sde.__dict__.get('initial_noise', _UNSET), sde.__dict__.get('noise_source', _UNSET), sde.__dict__.get('use_graph', _UNSET))
    try:
        sde.initial_noise = sharded_initial_noise(batch, event, seed, rank, world_size)
        if corrections > 0:
            sde.noise_source = KeyedNoise((lo, hi), event, seed + 1, corrections, sde.device.device)
        for m in coupled:
            m.shard = (lo, hi, batch, dist.group.WORLD if world_size > 1 else None)
        if coupled and world_size > 1:
            sde.use_graph = False            # (a collective per evaluation: the step is launched eagerly, not replayed)
        local = sde.sample((hi - lo,), c=c, steps=steps, corrections=corrections, tau=tau)
    finally:
        for name, value in zip(('initial_noise', 'noise_source', 'use_graph'), keep):
            if value is _UNSET:
                sde.__dict__.pop(name, None)     # (was the class default: leave no instance attribute behind)
            else:
                setattr(sde, name, value)
        for m in coupled:
            m.shard = None
    return all_gather_samples(local, batch) if gather else local


def _batch_coupled(sde) -> list:
    """The score modules under ``sde`` whose evaluation couples the samples of a batch (``DPSGaussianScore``)."""
    from .score import DPSGaussianScore
    return [m for m in sde.modules() if isinstance(m, DPSGaussianScore)]
