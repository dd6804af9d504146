#!/bin/bash
# where conv_wino4's (and conv_h2's) fabric traffic beyond the algorithmic bytes comes from.  L2 (TCC) request / hit / miss
# and fabric-side (EA) read / write request counters of tools/w4_traffic.py's layer launches, one rocprofv3 pass per counter group
# (counters alone: --kernel-trace only).   usage: tools/tcc_traffic.sh <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$1"; case "$OUT" in /*) ;; *) OUT="$R/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u > "$OUT/tcc_counters_available.txt"
run() { mode=$1; name=$2; shift 2
  SDA_MULTIPLY=$mode rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/${mode}_$name" -o p -- python $R/tools/w4_traffic.py > "$OUT/${mode}_$name.log" 2>&1
}
for mode in f32 f16x2; do
  run $mode req TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
  run $mode rw TCC_READ_sum TCC_WRITE_sum
  run $mode eard TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  run $mode eawr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  run $mode fetch FETCH_SIZE
  run $mode write WRITE_SIZE
done
cd $R
python tools/tcc_traffic_post.py "$OUT"
