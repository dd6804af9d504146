"""CPU, world_size 2 over gloo: the batch partition, the per-rank slices of the shared noise stream and the final
all-gather of sda_amd.parallel (the N>1 path of SURVEY 8e; the data path itself has no collective)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sda_amd import parallel as P


def test_shard_range_partitions():
    for total in (1, 7, 16, 128, 129):
        for ws in (1, 2, 3, 8):
            spans = [P.shard_range(total, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_initial_noise_is_a_slice_of_the_single_process_draw():
    full = P.sharded_initial_noise(7, (4, 3), 5, 0, 1)               # the single-process draw
    assert full.shape == (7, 4, 3)
    for ws in (2, 3, 8):
        parts = [P.sharded_initial_noise(7, (4, 3), 5, r, ws) for r in range(ws)]
        assert torch.equal(torch.cat(parts), full)
    assert not torch.equal(full, P.sharded_initial_noise(7, (4, 3), 6, 0, 1))


def _worker(rank, ws, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        batch, event = 5, (3, 2)
        lo, hi = P.shard_range(batch, rank, ws)
        # stand-in for the local sampler result: something every rank can recompute for checking
        full = torch.arange(batch * 6, dtype=torch.float32).reshape(batch, *event)
        gathered = P.all_gather_samples(full[lo:hi].clone(), batch)
        assert torch.equal(gathered, full)
        noise = P.ShardedNoise(batch, event, 11, rank, ws, 'cpu')
        z0, z1 = noise(0, 0), noise(0, 1)
        ref = torch.Generator().manual_seed(11)
        f0, f1 = torch.randn((batch,) + event, generator=ref), torch.randn((batch,) + event, generator=ref)
        assert torch.equal(z0, f0[lo:hi]) and torch.equal(z1, f1[lo:hi])
        assert P.world() == (rank, ws)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_gloo_world2_gather_and_noise():
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)
