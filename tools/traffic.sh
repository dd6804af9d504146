#!/bin/bash
# HBM-traffic evidence for conv_wino4 (run on the GPU box; results under gpurun_out/<tag>/):
#   1. FETCH_SIZE / WRITE_SIZE calibration on known byte counts in the kernel's own access shapes (tools/fetch_calib.hip)
#   2. per-layer FETCH_SIZE / WRITE_SIZE of conv_wino4 with the contiguous (SDA_W4_WALK=0) and the interleaved tile walk
#   3. the layer bench with both walks (HIP events; separate processes, no counters)
# usage: tools/traffic.sh <tag> [skip-bench]
set -u
TAG="$1"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o /tmp/fetch_calib || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o p -- /tmp/fetch_calib > $OUT/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o p -- /tmp/fetch_calib > $OUT/calib_write.log 2>&1
for walk in 0 1; do
  export SDA_W4_WALK=$walk
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/w4t_walk${walk}_fetch -o p -- python $R/tools/w4_traffic.py > $OUT/w4t_walk${walk}_fetch.log 2>&1
done
export SDA_W4_WALK=1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w4t_walk1_write -o p -- python $R/tools/w4_traffic.py > $OUT/w4t_walk1_write.log 2>&1
cd $R
python tools/w4_traffic_post.py $OUT $OUT/traffic.json > $OUT/traffic.txt 2>&1
cat $OUT/traffic.txt
if [ "${2:-}" != "skip-bench" ]; then
  for walk in 0 1; do
    SDA_W4_WALK=$walk python tools/wino4_check.py --skip-check --bench > $OUT/layer_bench_walk$walk.txt 2>&1
  done
  paste -d'\n' $OUT/layer_bench_walk0.txt $OUT/layer_bench_walk1.txt | grep "S=256\|S= 64" | cut -c1-110
fi
