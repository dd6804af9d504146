#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4h
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4h/trace -o t -- python $R/bench.py --workload lorenz_eval --lorenz-net local --no-cpu-baseline --graph 0 > $R/gpurun_out/r4h/bench.json 2> $R/gpurun_out/r4h/err.txt
cd $R; python tools/rocpd_summary.py $(ls gpurun_out/r4h/trace/*.db | head -1) | head -30
