mkdir -p gpurun_out/r06p
timeout 2400 python -m pytest tests/test_gpu_h2.py -x -q > gpurun_out/r06p/h2_tests.log 2>&1; echo "h2 tests rc=$?"; tail -5 gpurun_out/r06p/h2_tests.log
SDA_MULTIPLY=f16x2 timeout 1200 python tests/fuzz/h2_fuzz.py --cases 300 --seed 5 > gpurun_out/r06p/h2_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -4 gpurun_out/r06p/h2_fuzz.txt
for wl in kolmogorov64_default kolmogorov64; do python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline --other-configs 0 > gpurun_out/r06p/bench_$wl.json 2> gpurun_out/r06p/bench_$wl.err; python -c "
import json
d=json.loads(open('gpurun_out/r06p/bench_$wl.json').read().strip().splitlines()[-1])
print('$wl', d['ms_per_step'], d['opt_in_f16x2'])"; done
