mkdir -p gpurun_out/r06q
timeout 900 python tools/wino4_check.py --cases 200 > gpurun_out/r06q/wino4_check.txt 2>&1; echo "w4 rc=$?"; tail -1 gpurun_out/r06q/wino4_check.txt; grep FAIL gpurun_out/r06q/wino4_check.txt | head -5
timeout 900 python -m pytest tests/test_gpu_wino4_bm64.py -x -q 2>&1 | tail -2
W4Q_BM64=1 python tools/w4_quick_bench.py 2>/dev/null | tail -1
python bench.py --workload kolmogorov64_default --steps 5 --warmup 1 --second-line 0 --no-cpu-baseline --other-configs 0 > gpurun_out/r06q/bench_k64default.json 2>/dev/null
python bench.py --workload kolmogorov64 --steps 5 --warmup 1 --second-line 0 --no-cpu-baseline --other-configs 0 > gpurun_out/r06q/bench_k64.json 2>/dev/null
python -c "
import json
for f in ('bench_k64default','bench_k64'):
    d=json.loads(open('gpurun_out/r06q/'+f+'.json').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['roofline']['all_conv_algorithmic_tflops'], {k:(round(v['share_of_step'],3), round(v.get('mfma_util',0) or 0,3)) for k,v in d['roofline']['families'].items()})
"
