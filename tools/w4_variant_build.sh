#!/bin/bash
# A tooling build that differs from the product library in conv_wino4.hip's compile flags only:
#   tools/w4_variant_build.sh NAME -DFLAG=VALUE ...   ->  sda_amd/lib_NAME/libsda_hip.so  (the other objects are the product build's)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/sda_amd/lib_$NAME; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $R/sda_amd/csrc/conv_wino4.hip -o $D/conv_wino4.o
OBJS=$(ls $R/sda_amd/lib/*.o | grep -v conv_wino4.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libsda_hip.so $OBJS $D/conv_wino4.o
echo $D/libsda_hip.so
