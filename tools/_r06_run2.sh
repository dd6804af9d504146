mkdir -p gpurun_out/r06b
timeout 900 python tools/wino4_check.py --cases 250 > gpurun_out/r06b/wino4_check.txt 2>&1
echo "wino4_check rc=$?"
tail -30 gpurun_out/r06b/wino4_check.txt
python tools/conv_bench.py --widths 64,128,256 --n 960 --cin 7 > gpurun_out/r06b/convbench_default_widths.txt 2>&1
cat gpurun_out/r06b/convbench_default_widths.txt
python bench.py --workload kolmogorov64_default --steps 5 --warmup 1 --second-line 0 --no-cpu-baseline > gpurun_out/r06b/bench_k64default.json 2> gpurun_out/r06b/bench_k64default.err
tail -3 gpurun_out/r06b/bench_k64default.err
python -c "
import json
d=json.loads(open('gpurun_out/r06b/bench_k64default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['samples_finite'], d['roofline']['all_conv_algorithmic_tflops'])
for k,v in d['roofline']['families'].items(): print(k, round(v['share_of_step'],3), round(v['ms_per_step'],2), round(v.get('algorithmic_tflops',0),1), round(v.get('mfma_util',0) or 0,3))
"
