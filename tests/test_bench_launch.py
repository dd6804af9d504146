"""bench.py's launch path: `python bench.py --gpus N` -- the driver's verbatim call -- must start its own N
ranks instead of dying on a WORLD_SIZE assertion.  The reference has no launcher to mirror (experiments/lorenz/eval.py:42)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_plan_argument_plumbing():
    import bench
    # already a rank (torchrun / the driver's `python -m torch.distributed.run` form), or a single-GPU job: run in place
    assert bench.launch_plan(8, {'WORLD_SIZE': '8'}, ['--gpus', '8']) is None
    assert bench.launch_plan(1, {}, []) is None
    argv = ['--gpus', '4', '--steps', '20', '--warmup', '5', '--scaling', 'strong', '--workload', 'kolmogorov64']
    plan = bench.launch_plan(4, {'MASTER_PORT': '29777'}, argv)
    assert plan[0] == sys.executable and plan[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in plan and '--nproc-per-node=4' in plan
    assert plan[plan.index('--master-addr') + 1] == '127.0.0.1' and plan[plan.index('--master-port') + 1] == '29777'
    script = plan.index(os.path.join(ROOT, 'bench.py'))
    assert plan[script + 1:] == argv                       # every bench argument reaches the ranks unchanged
    # no MASTER_PORT in the environment: a free port is picked
    port = int(bench.launch_plan(2, {}, [])[bench.launch_plan(2, {}, []).index('--master-port') + 1])
    assert 1024 <= port < 65536


def test_too_few_gpus_for_rccl_is_a_clear_error_not_an_assertion():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8'], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0
    assert 'needs 8 visible GPUs' in p.stderr and 'AssertionError' not in p.stderr


def test_gpus8_strong_partition_and_row_keyed_inputs():
    """`--gpus 8 --scaling strong` (what the driver's SCALE run asks of configs[3]): the global batch is the configuration's 128
    trajectories, 16 per rank, and every rank's rows of y / x(1) are the rows the single-rank job draws (global-row keyed), so the
    N = 1, 2, 4, 8 points of the curve sample the SAME 128 trajectories.  Checked on the Lorenz-96 event (tiny rows); the Kolmogorov
    partition arithmetic on its own."""
    import torch
    import bench
    k = bench.WORKLOADS['kolmogorov256']
    assert [bench.partition(k['per_gpu'], 'strong', r, 8) for r in range(8)] == [(16, 128, 16 * r) for r in range(8)]
    assert bench.partition(k['per_gpu'], 'strong', 0, 1) == (128, 128, 0)
    assert [bench.partition(k['per_gpu'], 'strong', r, 4)[0] for r in range(4)] == [32] * 4
    assert [bench.partition(k['per_gpu'], 'weak', r, 8) for r in range(8)] == [(16, 128, 16 * r) for r in range(8)]
    assert bench.partition(k['per_gpu'], 'weak', 0, 1) == (16, 16, 0)
    wl = dict(bench.WORKLOADS['lorenz96'])
    wl['per_gpu'] = 2                                   # global batch 16
    event = (wl['L'], wl['state'])
    y1, x1 = bench.rank_inputs(wl, event, 'strong', 0, 1)
    assert x1.shape == (16,) + event and y1.shape == (16, 16, 1)
    for world in (2, 4, 8):
        parts = [bench.rank_inputs(wl, event, 'strong', r, world) for r in range(world)]
        assert torch.equal(torch.cat([p[0] for p in parts]), y1) and torch.equal(torch.cat([p[1] for p in parts]), x1)
    # weak scaling: rank r of an N-rank job draws the rows r * per_gpu ... of one global stream as well
    w8 = [bench.rank_inputs(wl, event, 'weak', r, 8) for r in range(8)]
    w1 = bench.rank_inputs(wl, event, 'weak', 0, 1)
    assert torch.equal(w8[0][1], w1[1]) and torch.equal(w8[0][0], w1[0])
    assert not torch.equal(w8[1][1], w8[0][1])
    plan = bench.launch_plan(8, {'MASTER_PORT': '29555'}, ['--gpus', '8', '--scaling', 'strong', '--steps', '5', '--warmup', '2'])
    assert '--nproc-per-node=8' in plan and plan[-8:] == ['--gpus', '8', '--scaling', 'strong', '--steps', '5', '--warmup', '2']
