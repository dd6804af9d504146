"""GPU: the pieces a batch-sharded (one process per GPU) posterior-sampling job runs on each rank -- the row-keyed noise
kernel (sda_randn_rows) against its numpy restatement, and `parallel.sample_sharded` itself, with ranks 0 and 1 of a 2-rank
job emulated one after the other on this GPU: their shards concatenate to the single-rank job bit for bit (guided, one
Langevin correction per step), eagerly and under hipGraph replay.  (The collective itself -- one all-gather after the
loop -- is covered by the gloo world-2 CPU test; no multi-GPU node is available to these tests.)"""
import ctypes

import numpy as np
import pytest
import torch

from tests import philox_ref
from tests.util import build_mcscore2d_tiny, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def test_philox_words_bit_exact(dev):
    from sda_amd import _lib
    lib = _lib.load()
    n = 4099
    for seed, c1, c2, c3 in ((0, 0, 0, 0), (0x299f31d0a4093822, 0x85a308d3, 0x13198a2e, 0x03707344), (2 ** 64 - 1, 7, 2 ** 32 - 1, 5)):
        out = torch.empty(4 * n, dtype=torch.int32, device=dev)
        _lib.check(lib.sda_philox_words(out.data_ptr(), n, seed, c1, c2, c3, torch.cuda.current_stream().cuda_stream), 'philox')
        got = out.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, philox_ref.philox_words(n, seed, c1, c2, c3))
    # Random123 known-answer vector (counter = key = 0)
    assert [hex(int(v)) for v in philox_ref.philox_words(1, 0, 0, 0, 0)] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']


@pytest.mark.parametrize('rows,per_row,row0,draw', [(3, 64, 0, 0), (2, 1001, 5, 7), (1, 7, 2 ** 33 + 1, 2 ** 35 + 3), (5, 4096, 120, 1999)])
def test_randn_rows_matches_numpy_restatement(dev, rows, per_row, row0, draw):
    from sda_amd import ops
    out = torch.empty(rows, per_row, device=dev)
    ops.randn_rows(out, 0x1234567890abcdef, row0, draw=draw)
    ref = philox_ref.randn_rows(rows, per_row, 0x1234567890abcdef, row0, draw)
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-5        # fp32 log / sincos round-off on |z| <= 5.8
    # the device-side draw index gives the same tensor
    d = torch.tensor([draw // 3], device=dev, dtype=torch.int64)
    out2 = torch.empty_like(out)
    ops.randn_rows(out2, 0x1234567890abcdef, row0, draw_dev=d, draw_mul=3, draw_add=draw % 3)
    assert torch.equal(out, out2)


def test_randn_rows_is_standard_normal_and_row_keyed(dev):
    from scipy import stats
    from sda_amd import ops
    out = torch.empty(8, 1 << 18, device=dev)
    ops.randn_rows(out, 42, 16, draw=3)
    z = out.cpu().numpy().astype(np.float64)
    assert abs(z.mean()) < 3e-3 and abs(z.std() - 1) < 3e-3
    assert stats.kstest(z.reshape(-1)[:500000], 'norm').pvalue > 1e-3
    assert abs(np.corrcoef(z[0], z[1])[0, 1]) < 0.01 and abs(np.corrcoef(z[0][:-1], z[0][1:])[0, 1]) < 0.01
    # rows 2..5 generated alone (another rank's shard) are the same numbers
    part = torch.empty(4, 1 << 18, device=dev)
    ops.randn_rows(part, 42, 18, draw=3)
    assert torch.equal(part, out[2:6])
    other = torch.empty(4, 1 << 18, device=dev)
    ops.randn_rows(other, 42, 18, draw=4)
    assert not torch.equal(other, part)


def _guided_sde(dev, net=None, event=(5, 2, 8, 8)):
    from sda_amd import observe as Ob
    from sda_amd.score import GaussianScore, VPSDE
    g, grp = load_golden('mcscore2d_tiny')
    if net is None:
        net = build_mcscore2d_tiny()
        net.load_state_dict(grp['sd'])
    torch.manual_seed(0)
    A = Ob.Subsample((slice(None, None, 2), slice(None), slice(None, None, 2), slice(None, None, 2)))
    y = torch.randn(A(torch.empty((1,) + event, device=dev)).shape)
    gs = GaussianScore(y, A=A, std=0.5, sde=VPSDE(net, shape=()), gamma=1e-2)
    return VPSDE(gs, shape=event).to(dev)


@pytest.mark.parametrize('use_graph', [False, True])
def test_sample_sharded_two_emulated_ranks_equal_single_rank(dev, use_graph):
    from sda_amd import parallel as P
    sde = _guided_sde(dev)
    sde.use_graph = use_graph
    batch, kw = 5, dict(steps=6, corrections=1, tau=0.5, seed=3)
    whole = P.sample_sharded(sde, batch, rank=0, world_size=1, **kw)
    assert whole.shape == (batch, 5, 2, 8, 8) and torch.isfinite(whole).all()
    for ws in (2, 3):
        shards = [P.sample_sharded(sde, batch, rank=r, world_size=ws, **kw) for r in range(ws)]
        assert [s.shape[0] for s in shards] == [hi - lo for lo, hi in (P.shard_range(batch, r, ws) for r in range(ws))]
        assert torch.equal(torch.cat(shards), whole), f'world {ws}: shards differ from the single-rank run'
    again = P.sample_sharded(sde, batch, rank=0, world_size=1, **kw)
    assert torch.equal(again, whole)
    assert sde.noise_source is None and sde.initial_noise is None


def test_sample_sharded_graph_equals_eager(dev):
    from sda_amd import parallel as P
    sde = _guided_sde(dev)
    kw = dict(steps=5, corrections=2, tau=0.3, seed=8)
    eager = P.sample_sharded(sde, 4, rank=1, world_size=2, **kw)
    sde.use_graph = True
    graph = P.sample_sharded(sde, 4, rank=1, world_size=2, **kw)
    assert torch.allclose(graph, eager, rtol=1e-5, atol=1e-6)


def test_midsize_sharded_rows_are_independent_of_their_group(dev):
    """K64-style mid-size net, 3 trajectories: rank shards of sizes (2, 1) vs the single-rank run."""
    from sda_amd import parallel as P
    from sda_amd.experiments.kolmogorov import make_score
    torch.manual_seed(1)
    net = make_score(window=5, embedding=32, hidden_channels=(16, 32, 64), hidden_blocks=(1, 1, 1), size=16)
    sde = _guided_sde(dev, net, event=(7, 2, 16, 16))
    kw = dict(steps=3, corrections=1, tau=0.5, seed=2)
    whole = P.sample_sharded(sde, 3, rank=0, world_size=1, **kw)
    shards = [P.sample_sharded(sde, 3, rank=r, world_size=2, **kw) for r in range(2)]
    assert torch.equal(torch.cat(shards), whole)


def _dps_sde(dev, event=(5, 2, 8, 8), fused_adjoint=True):
    from sda_amd import observe as Ob
    from sda_amd.score import DPSGaussianScore, VPSDE
    g, grp = load_golden('mcscore2d_tiny')
    net = build_mcscore2d_tiny()
    net.load_state_dict(grp['sd'])
    torch.manual_seed(0)
    sub = Ob.Subsample((slice(None, None, 2), slice(None), slice(None, None, 2), slice(None, None, 2)))
    A = sub if fused_adjoint else (lambda x: x[..., ::2, :, ::2, ::2])     # hand-written adjoint / autograd through A
    y = torch.randn(sub(torch.empty((1,) + event, device=dev)).shape)
    return VPSDE(DPSGaussianScore(y, A=A, sde=VPSDE(net, shape=()), zeta=0.7), shape=event).to(dev)


def _dps_rank(rank, ws, port, ret):
    import os
    import torch.distributed as dist
    from sda_amd import parallel as P
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=ws)      # (both ranks on cuda:0; RCCL wants one GPU per rank)
    try:
        dev = torch.device('cuda:0')
        kw = dict(steps=4, corrections=1, tau=0.5, seed=6)
        worst = 0.0
        for fused in (True, False):
            sde = _dps_sde(dev, fused_adjoint=fused)
            got = P.sample_sharded(sde, 5, **kw)                   # one scalar all-reduce per score evaluation + the gather
            alone = P.sample_sharded(sde, 5, rank=0, world_size=1, **kw)
            assert got.shape == alone.shape == (5, 5, 2, 8, 8) and torch.isfinite(alone).all()
            worst = max(worst, ((got - alone).abs().max() / alone.abs().max()).item())
            lo, hi = P.shard_range(5, rank, ws)
            sde.initial_noise = P.sharded_initial_noise(5, (5, 2, 8, 8), 6, rank, ws)
            sde.noise_source = P.KeyedNoise((lo, hi), (5, 2, 8, 8), 7, 1, dev)
            replica = sde.sample((hi - lo,), steps=4, corrections=1, tau=0.5)      # uncoupled: its own shard's error norm
            sde.initial_noise = sde.noise_source = None
            assert ((replica - alone[lo:hi]).abs().max() / alone.abs().max()).item() > 1e-3
        ret[rank] = worst
    finally:
        dist.destroy_process_group()


def test_dps_sharded_over_two_ranks_equals_single_process(dev):
    """DPSGaussianScore's error norm is a sum over the WHOLE batch (sda/score.py:339-342): the one exchange step on the path.
    Two ranks (gloo, both on this GPU) all-reduce that scalar in every evaluation and reproduce the single-process run;
    two uncoupled replicas do not."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() * 7 + 3) % 2000
    procs = [ctx.Process(target=_dps_rank, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert ret[0] <= 2e-5 and ret[1] <= 2e-5, dict(ret)
