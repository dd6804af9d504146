#!/bin/bash
# conv_h2 ablation ON ZEROS (no power-managed clock drop: what each part costs in CYCLES).  Prebuilt tooling library as in h2_ablate_prebuilt.sh.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SDA_HIP_LIB=$R/sda_amd/lib_abl/libsda_hip.so SDA_MULTIPLY=f16x2
for abl in 0 1 2 3 4 8 11; do
  echo "== SDA_H2_ABL=$abl"
  SDA_H2_ABL=$abl python $R/tools/h2_zero_probe.py --zeros 2>/dev/null | cut -c1-60
done
for stg in 1 2 3; do
  echo "== stagger $stg (x 8 128 cycles per slot phase), full kernel, zeros then random"
  SDA_H2_STAGGER=$stg python $R/tools/h2_zero_probe.py --zeros 2>/dev/null | cut -c1-60
  SDA_H2_STAGGER=$stg python $R/tools/h2_zero_probe.py 2>/dev/null | grep random | cut -c1-60
done
