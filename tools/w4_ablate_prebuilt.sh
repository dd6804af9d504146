#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SDA_HIP_LIB=$R/sda_amd/lib_abl/libsda_hip.so
for dbg in 0 16 256 272 32 64 1024 1040; do
  echo "== SDA_CONV_DEBUG=$dbg"
  SDA_CONV_DEBUG=$dbg python $R/tools/w4_quick_bench.py 2>/dev/null | tr '|' '\n' | head -6
done
