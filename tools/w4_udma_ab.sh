#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== correctness (product build: U slab by LDS-DMA)"
timeout 600 python $R/tools/wino4_check.py --cases 120 2>&1 | grep -v amdgpu.ids | tail -6
for rep in 1 2; do
echo "== timing, product (UDMA)"; python $R/tools/w4_quick_bench.py 2>/dev/null | tr '|' '\n'
echo "== timing, -DW4_UDMA=0 (registers + ds_write)"; SDA_HIP_LIB=$R/sda_amd/lib_abl/libsda_hip.so python $R/tools/w4_quick_bench.py 2>/dev/null | tr '|' '\n'
done
