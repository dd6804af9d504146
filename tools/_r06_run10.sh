mkdir -p gpurun_out/r06k
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_fused1d.py tests/test_gpu_lorenz_eval.py tests/test_gpu_configs.py -x -q -k "1d or lorenz or fused or whole_net or config0 or config1" > gpurun_out/r06k/tests1d.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r06k/tests1d.log
python tools/net1d_trace.py 2>&1 | grep -v amdgpu > gpurun_out/r06k/net1d_trace.txt; head -24 gpurun_out/r06k/net1d_trace.txt
for wl in lorenz96 lorenz63; do python bench.py --workload $wl --steps 2000 --warmup 200 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'], d['ms_per_step'])"; done
