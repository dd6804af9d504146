mkdir -p gpurun_out/r06d
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -k "stride2_vjp" > gpurun_out/r06d/par4_tests.log 2>&1
echo "par4 rc=$?"; tail -8 gpurun_out/r06d/par4_tests.log
python bench.py --workload kolmogorov64_default --steps 5 --warmup 1 --second-line 0 --no-cpu-baseline > gpurun_out/r06d/bench_k64default.json 2> gpurun_out/r06d/bench_k64default.err
tail -3 gpurun_out/r06d/bench_k64default.err
python -c "
import json
d=json.loads(open('gpurun_out/r06d/bench_k64default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['samples_finite'], d['roofline']['all_conv_algorithmic_tflops'])
for k,v in d['roofline']['families'].items(): print(k, round(v['share_of_step'],3), round(v['ms_per_step'],2), round(v.get('algorithmic_tflops',0),1), round(v.get('mfma_util',0) or 0,3), v['launches_per_step'])
"
