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
