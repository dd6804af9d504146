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


def _tiny_guided_sde():
    from sda_amd.score import GaussianScore, VPSDE
    from tests.util import build_mcscore2d_tiny, load_golden
    g, grp = load_golden('mcscore2d_tiny')
    net = build_mcscore2d_tiny()
    net.load_state_dict(grp['sd'])
    A = lambda x: x[..., ::2, :, ::2, ::2]
    gs = GaussianScore(g['y_obs'][:1], A=A, std=0.5, sde=VPSDE(net, shape=()), gamma=1e-2)     # one shared observation
    return VPSDE(gs, shape=(5, 2, 8, 8))


def _tiny_dps_sde(per_sample_y: bool, batch: int):
    """DPSGaussianScore: its error norm is summed over the WHOLE batch (score.py:339) -- the one collective on the path."""
    from sda_amd.score import DPSGaussianScore, VPSDE
    from tests.util import build_mcscore2d_tiny, load_golden
    g, grp = load_golden('mcscore2d_tiny')
    net = build_mcscore2d_tiny()
    net.load_state_dict(grp['sd'])
    A = lambda x: x[..., ::2, :, ::2, ::2]
    y = g['y_obs'][:1]
    if per_sample_y:                                   # one observation per trajectory: follows the rows of a shard
        y = torch.cat([y * (1 + 0.25 * i) for i in range(batch)])
    return VPSDE(DPSGaussianScore(y, A=A, sde=VPSDE(net, shape=()), zeta=0.7), shape=(5, 2, 8, 8))


def _worker(rank, ws, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    mpatch = pytest.MonkeyPatch()
    try:
        batch, event = 5, (3, 2)
        lo, hi = P.shard_range(batch, rank, ws)
        # stand-in for the local sampler result: something every rank can recompute for checking
        full = torch.arange(batch * 6, dtype=torch.float32).reshape(batch, *event)
        gathered = P.all_gather_samples(full[lo:hi].clone(), batch)
        assert torch.equal(gathered, full)
        # equal shares (every BASELINE configuration): the single all_gather_into_tensor path
        full6 = torch.arange(6 * 6, dtype=torch.float32).reshape(6, *event)
        lo6, hi6 = P.shard_range(6, rank, ws)
        assert torch.equal(P.all_gather_samples(full6[lo6:hi6].clone(), 6), full6)
        assert P.world() == (rank, ws)
        # the whole sharded sampler (Python orchestration on the test-only CPU shim; kernels are covered by -m gpu):
        # guided, one Langevin correction per step; the gathered result equals the single-process run of the same job
        from tests import cpu_shim
        cpu_shim.install(mpatch)
        sde = _tiny_guided_sde()
        got = P.sample_sharded(sde, 3, steps=2, corrections=1, tau=0.5, seed=4)
        assert got.shape == (3, 5, 2, 8, 8)
        alone = P.sample_sharded(sde, 3, steps=2, corrections=1, tau=0.5, seed=4, rank=0, world_size=1)
        assert torch.equal(got, alone), (got - alone).abs().max()
        mine = P.sample_sharded(sde, 3, steps=2, corrections=1, tau=0.5, seed=4, gather=False)
        assert torch.equal(mine, alone[P.shard_range(3, rank, ws)[0]:P.shard_range(3, rank, ws)[1]])
        # DPS guidance couples the batch through one scalar: sharded = one all-reduce per evaluation, and the result still
        # equals the single-process run (up to the summation order of that scalar)
        for per_sample in (False, True):
            dps = _tiny_dps_sde(per_sample, 3)
            got = P.sample_sharded(dps, 3, steps=2, corrections=1, tau=0.5, seed=4)
            alone = P.sample_sharded(dps, 3, steps=2, corrections=1, tau=0.5, seed=4, rank=0, world_size=1)
            assert got.shape == alone.shape == (3, 5, 2, 8, 8)
            assert all(m.shard is None for m in P._batch_coupled(dps))
            close = lambda a, b: (a - b).abs().max() <= 2e-5 * b.abs().max()        # (max-norm: 2 steps from t = 1 are large)
            assert close(got, alone), ((got - alone).abs().max(), alone.abs().max())
            # ... and NOT the result of two uncoupled replicas (each normalising by its own shard's error)
            if per_sample:
                continue
            lo, hi = P.shard_range(3, rank, ws)
            dps.initial_noise = P.sharded_initial_noise(3, (5, 2, 8, 8), 4, rank, ws)
            dps.noise_source = P.KeyedNoise((lo, hi), (5, 2, 8, 8), 5, 1, 'cpu')
            replica = dps.sample((hi - lo,), steps=2, corrections=1, tau=0.5)
            dps.initial_noise = dps.noise_source = None
            assert not close(replica, alone[lo:hi])
        with pytest.raises(ValueError, match='emulated rank'):
            P.sample_sharded(_tiny_dps_sde(False, 3), 3, steps=1, rank=0, world_size=2)
        ret[rank] = True
    finally:
        mpatch.undo()
        dist.destroy_process_group()


def test_keyed_noise_rows_do_not_depend_on_the_partition(monkeypatch):
    from tests import cpu_shim
    cpu_shim.install(monkeypatch)
    event = (4, 3)
    whole = P.KeyedNoise((0, 7), event, 9, 2, 'cpu')
    for ws in (2, 3):
        for step, corr in ((0, 0), (3, 1)):
            parts = [P.KeyedNoise(P.shard_range(7, r, ws), event, 9, 2, 'cpu')(step, corr) for r in range(ws)]
            assert torch.equal(torch.cat(parts), whole(step, corr))
    assert not torch.equal(whole(0, 0), whole(0, 1))
    assert not torch.equal(whole(0, 1), whole(1, 0))
    # device-side draw index (what a replayed hipGraph reads) == host-side draw index
    assert torch.equal(whole.draw_dev(torch.tensor([3]), 1), whole(3, 1))
    z = P.KeyedNoise((0, 64), (1000,), 1, 1, 'cpu')(0, 0)
    assert abs(z.mean().item()) < 0.02 and abs(z.std().item() - 1) < 0.02


def test_gloo_world2_sample_sharded_equals_single_process():
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)


def _worker8(rank, ws, port, ret):
    """One of 8 ranks of the real partition shape scaled down (configs[3]: 128 trajectories -> 16 per rank; here 16 -> 2 per rank):
    equal shares, i.e. the single `all_gather_into_tensor` path."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    mpatch = pytest.MonkeyPatch()
    try:
        from tests import cpu_shim
        cpu_shim.install(mpatch)
        calls = []
        real = dist.all_gather_into_tensor
        mpatch.setattr(dist, 'all_gather_into_tensor', lambda out, inp, **kw: (calls.append(tuple(out.shape)), real(out, inp, **kw))[1])
        batch = 2 * ws
        assert P.shard_range(batch, rank, ws) == (2 * rank, 2 * rank + 2)
        sde = _tiny_guided_sde()
        got = P.sample_sharded(sde, batch, steps=2, corrections=1, tau=0.5, seed=7)
        assert got.shape == (batch, 5, 2, 8, 8) and calls == [(batch, 5, 2, 8, 8)]
        assert sde.initial_noise is None and sde.noise_source is None
        if rank == 0:                                   # the single-process run of the same job, once
            alone = P.sample_sharded(sde, batch, steps=2, corrections=1, tau=0.5, seed=7, rank=0, world_size=1)
            ret['equal'] = bool(torch.equal(got, alone))
            ret['maxdiff'] = float((got - alone).abs().max())
        ret[rank] = True
    finally:
        mpatch.undo()
        dist.destroy_process_group()


def test_gloo_world8_equal_shares_equals_single_process():
    """world_size 8 -- the node the BASELINE configurations are quoted on: 8 ranks x 2 trajectories gathered by ONE
    all_gather_into_tensor equal the 16-trajectory single-process run bit for bit (row-keyed noise, no in-loop collective)."""
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, ret)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(8))
    assert ret['equal'], ret['maxdiff']


def test_sample_sharded_restores_what_the_caller_had_set(monkeypatch):
    from tests import cpu_shim
    cpu_shim.install(monkeypatch)
    sde = _tiny_guided_sde()
    mine = lambda i, j: torch.zeros(1, 5, 2, 8, 8)
    sde.noise_source = mine
    sde.use_graph = False
    P.sample_sharded(sde, 2, steps=1, corrections=1, tau=0.5, seed=1, rank=0, world_size=1)
    assert sde.noise_source is mine and sde.initial_noise is None and sde.__dict__['use_graph'] is False
    fresh = _tiny_guided_sde()
    with pytest.raises(ZeroDivisionError):
        monkeypatch.setattr(P, 'KeyedNoise', lambda *a, **k: 1 / 0)
        P.sample_sharded(fresh, 2, steps=1, corrections=1, tau=0.5, seed=1, rank=0, world_size=1)
    assert fresh.initial_noise is None and 'initial_noise' not in fresh.__dict__ and 'use_graph' not in fresh.__dict__
