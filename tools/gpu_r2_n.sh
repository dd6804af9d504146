#!/bin/bash
mkdir -p gpurun_out/r2n
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2n/l96b -o t -- python $GRAFT_REPO_ROOT/bench.py --workload lorenz96 --steps 20 --warmup 2 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(ls gpurun_out/r2n/l96b/*.db | head -1) | head -45
