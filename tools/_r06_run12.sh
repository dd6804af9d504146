mkdir -p gpurun_out/r06n
timeout 1500 python -m pytest tests/test_gpu_wino4_bm64.py tests/test_gpu_ops.py -x -q -k "stride2_vjp or unet_default or default_width" > gpurun_out/r06n/tests.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/r06n/tests.log
python tools/conv_bench.py --widths 32,64,128 --n 960 --cin 7 --only "zi" 2>&1 | grep -v amdgpu
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06n/bench_driver_cmd.json 2> gpurun_out/r06n/bench_driver_cmd.err; tail -2 gpurun_out/r06n/bench_driver_cmd.err
python -c "
import json
d=json.loads(open('gpurun_out/r06n/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['opt_in_f16x2'].get('ms_per_step'), {k:(v.get('ms_per_step'), v.get('all_conv_algorithmic_tflops'), v.get('error')) for k,v in d['other_configs'].items() if isinstance(v,dict)}, d['other_configs']['total_wall_s'])
"
