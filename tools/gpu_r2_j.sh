#!/bin/bash
bash tools/profile_bench.sh r02_kolmogorov256_g1c1 --steps 1 --warmup 1 --profile-steps 0 2>&1 | tail -14
cat profiles/r02_kolmogorov256_g1c1_traffic.json
