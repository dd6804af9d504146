#!/bin/bash
# kernel traces of the Lorenz workloads (rocprofv3 --kernel-trace --stats), fused path
for wl in lorenz63 lorenz96; do
  PROFILE_PMC=0 PROFILE_KERNEL=net1d bash tools/profile_bench.sh r04a_${wl}_g1c1 --workload $wl --steps 200 --warmup 20 2>&1 | tail -14
done
