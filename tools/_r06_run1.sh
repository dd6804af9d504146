set -x
mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_rccl.py -x -q > gpurun_out/r06a/rccl.log 2>&1
python tools/conv_bench.py --widths 64,128,256 --n 960 --cin 7 > gpurun_out/r06a/convbench_default_widths.txt 2>&1
python tools/conv_bench.py --n 896 > gpurun_out/r06a/convbench_k64.txt 2>&1
python bench.py --workload kolmogorov64_default --steps 5 --warmup 1 --second-line 0 > gpurun_out/r06a/bench_k64default.json 2> gpurun_out/r06a/bench_k64default.err
python bench.py --workload kolmogorov64 --steps 5 --warmup 1 --second-line 0 --no-cpu-baseline > gpurun_out/r06a/bench_k64.json 2> gpurun_out/r06a/bench_k64.err
python bench.py > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err
tail -5 gpurun_out/r06a/rccl.log
