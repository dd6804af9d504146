"""First contact with RCCL on a 1-GPU box: the 8-GPU job must not be the first time librccl, the HSA IPC setting
(HSA_ENABLE_IPC_MODE_LEGACY=0), `init_process_group('nccl', device_id=...)` and libsda_hip.so share a process.  One rank, live process
group, the collectives of the data-parallel path issued for real (`all_gather_samples(..., always_collective=True)`; the DPS scalar
all-reduce), and bench.py's own code path with the group alive (`--force-pg 1`).  The reference has no collective to mirror
(SURVEY.md section 5); the data-parallel split is SURVEY 8(e)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_RANK_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)          # nccl == RCCL on ROCm
from sda_amd import _lib, parallel
from sda_amd.experiments.lorenz import make_global_score
from sda_amd.score import DPSGaussianScore, GaussianScore, VPSDE
lib = _lib.load()                                        # libsda_hip.so beside librccl in one process
torch.manual_seed(0)
net = make_global_score(channels=3).to(dev)
y = torch.randn(8, 1)
A = lambda x: x[..., ::8, :1]
out = {'ranks_seen': dist.get_world_size(), 'backend': dist.get_backend(), 'hsa_ipc_legacy': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}
# 1. the job's one collective, issued through RCCL although a single rank would not need it
sde = VPSDE(GaussianScore(y, A=A, std=0.5, sde=VPSDE(net, shape=())), shape=(64, 3)).to(dev)
local = parallel.sample_sharded(sde, 6, steps=3, corrections=1, tau=0.5, seed=5, gather=False)
whole = parallel.all_gather_samples(local, 6, always_collective=True)
out['gather_equal'] = bool(torch.equal(whole, local)) and whole.data_ptr() != local.data_ptr()
out['gather_finite'] = bool(torch.isfinite(whole).all().item())
# 2. the one in-loop exchange of the path: DPSGaussianScore's batch-global scalar, all-reduced per evaluation (world 1: identity)
dps = VPSDE(DPSGaussianScore(y, A=A, sde=VPSDE(net, shape=()), zeta=0.5), shape=(64, 3)).to(dev)
m = [mm for mm in dps.modules() if isinstance(mm, DPSGaussianScore)][0]
ref = parallel.sample_sharded(dps, 4, steps=2, seed=7, gather=False)
m_shard_seen = []
orig = dist.all_reduce
def spy(t, *a, **k):
    m_shard_seen.append(tuple(t.shape))
    return orig(t, *a, **k)
dist.all_reduce = spy
try:
    lo_hi = (0, 4, 4, dist.group.WORLD)
    dps.use_graph = False
    dps.initial_noise = parallel.sharded_initial_noise(4, (64, 3), 7, 0, 1)
    m.shard = lo_hi
    got = dps.sample((4,), steps=2)
finally:
    dist.all_reduce = orig
    m.shard = None
out['dps_allreduce_calls'] = len(m_shard_seen)
out['dps_equal'] = float((got - ref).abs().max().item())
t = torch.ones(4, device=dev)
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
out['allreduce_ok'] = bool((t == 1).all().item())
dist.destroy_process_group()
print('RCCL1 ' + json.dumps(out), flush=True)
'''


def _free_port() -> str:
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return str(so.getsockname()[1])


def _env():
    env = dict(os.environ)
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port())
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return env


def test_rccl_process_group_of_one_rank_runs_the_collectives():
    p = subprocess.run([sys.executable, '-c', _RANK_SCRIPT, ROOT], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('RCCL1 ')]
    assert line, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(line[-1][6:])
    assert out['ranks_seen'] == 1 and out['backend'] == 'nccl'
    assert out['gather_equal'] and out['gather_finite'] and out['allreduce_ok']
    assert out['dps_allreduce_calls'] >= 2 and out['dps_equal'] < 1e-6, out


def test_bench_code_path_with_a_live_process_group():
    """`bench.py --gpus 1` with the group initialised: barrier + max-over-ranks all-reduce around the timed region and the final
    all-gather through RCCL; `ranks_seen` is read from the live group."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-pg', '1', '--workload', 'lorenz96',
                        '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--other-configs', '0'],
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    cfg = out['config']
    assert cfg['process_group_live'] is True and cfg['ranks_seen'] == 1 and cfg['backend'] == 'nccl (RCCL)'
    assert out['samples_finite'] and out['n_gpus'] == 1 and out['final_allgather_ms'] > 0
