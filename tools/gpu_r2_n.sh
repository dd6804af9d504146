#!/bin/bash
mkdir -p gpurun_out/r2n
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2n/l96 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload lorenz96 --steps 20 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r2n/l96.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2n/l96.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(ls gpurun_out/r2n/l96/*.db | head -1) | head -50
tail -1 gpurun_out/r2n/l96.json | cut -c1-300
