#!/bin/bash
# What each part of conv_h2's inner loop costs, measured by leaving it out (tooling build -DSDA_H2_ABLATE; results are WRONG in these runs).
#   bits: 1 no weight loads in the loop, 2 no loader, 4 no MFMAs, 8 no output stores
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SDA_LIBDIR=/tmp/sda_h2_abl SDA_EXTRA_HIPCC_FLAGS=-DSDA_H2_ABLATE
python -m sda_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
export SDA_HIP_LIB=/tmp/sda_h2_abl/libsda_hip.so SDA_MULTIPLY=f16x2
for abl in 0 1 2 3 4 7 8 15; do
  echo "== SDA_H2_ABL=$abl"
  SDA_H2_ABL=$abl python $R/tools/h2_check.py --timing-only --plain 2>/dev/null | grep -E "plain"
done
