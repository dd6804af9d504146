#!/bin/bash
# rocprofv3 evidence for one bench.py command line (run on the GPU box):
# (only gpurun_out/ travels back from the GPU box: run `python tools/profile_post.py <tag>` here to fill profiles/)
#   1. --kernel-trace --stats     -> profiles/<tag>_kernel_stats.csv   (per-kernel calls / total / average)
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, counters alone) -> profiles/<tag>_traffic.json
#      HBM bytes per convolution launch (conv_wino_kernel + conv_igemm*) = (2*FETCH_SIZE + WRITE_SIZE) * 1024   [gfx950: FETCH_SIZE reports half of a
#      wide coalesced read stream, MI355X_MICROARCH.md section HBM; counters are in KiB]
# usage: [PROFILE_KERNEL=net1d] [PROFILE_PMC=0] tools/profile_bench.sh <tag> <bench args...>     (PROFILE_PMC=0: kernel trace only)
set -u
TAG="$1"; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT $R/profiles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py "$@" --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
if [ "${PROFILE_PMC:-1}" != 0 ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-profile > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-profile > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/sq -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-profile > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-profile > /dev/null 2> $OUT/grbm.err
fi
cd $R
# (profiles/ written on the GPU box does not travel back: re-run this line here on the merged gpurun_out/)
python tools/profile_post.py "$TAG" ${PROFILE_KERNEL:+--kernel "$PROFILE_KERNEL"} | head -12
