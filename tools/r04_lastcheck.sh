mkdir -p gpurun_out/check
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for wl in qg128 lorenz96 lorenz63; do
  timeout 900 python bench.py --workload $wl --steps 50 --warmup 5 > gpurun_out/check/bench_$wl.json 2> /dev/null; tail -1 gpurun_out/check/bench_$wl.json | cut -c1-200
done
timeout 900 python bench.py --gpus 2 --backend gloo --workload kolmogorov64 --steps 4 --warmup 1 > gpurun_out/check/bench_2rank_gloo_kolmogorov64.json 2> gpurun_out/check/bench_2rank_gloo.err
tail -1 gpurun_out/check/bench_2rank_gloo_kolmogorov64.json | cut -c1-300
for net in global local; do timeout 900 python bench.py --workload lorenz_eval --lorenz-net $net --lorenz-freq lo --cpu-seconds 8 > gpurun_out/check/lorenz_eval_${net}_lo.json 2> /dev/null; tail -1 gpurun_out/check/lorenz_eval_${net}_lo.json | cut -c1-260; done
