#!/bin/bash
# conv_h2 ablation with a PREBUILT tooling library (built in the build container:
#   SDA_LIBDIR=sda_amd/lib_abl SDA_EXTRA_HIPCC_FLAGS=-DSDA_H2_ABLATE python -m sda_amd.build ) -- no compile time on the GPU box.
#   bits: 1 no weight loads in the loop, 2 no loader, 4 no MFMAs, 8 no output stores   (results are WRONG in these runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SDA_HIP_LIB=$R/sda_amd/lib_abl/libsda_hip.so SDA_MULTIPLY=f16x2
for abl in 0 1 2 3 4 7 8 15; do
  echo "== SDA_H2_ABL=$abl"
  SDA_H2_ABL=$abl python $R/tools/h2_check.py --timing-only --plain 2>/dev/null | grep -E "plain" | cut -c1-75
done
