mkdir -p gpurun_out/r06i gpurun_out/profiles_r06
prof() {  # tag, kernel, bench args...: the passes, the summaries into gpurun_out/profiles_r06, the raw rocprofv3 output dropped (64 MiB limit)
  tag=$1; shift; kern=$1; shift
  bash tools/profile_bench.sh $tag "$@" > gpurun_out/r06i/$tag.log 2>&1
  python tools/profile_post.py $tag --kernel $kern --dst gpurun_out/profiles_r06 >> gpurun_out/r06i/$tag.log 2>&1
  for p in trace fetch write sq grbm; do tail -2 gpurun_out/prof_$tag/$p.err >> gpurun_out/r06i/$tag.log 2>/dev/null; done
  rm -rf gpurun_out/prof_$tag
}
prof r06_kolmogorov256_g1c1 conv_wino4_kernel --steps 2 --warmup 1 --second-line 0 --other-configs 0
prof r06_kolmogorov64_default_g1c1 conv_wino4_kernel --workload kolmogorov64_default --steps 4 --warmup 1 --second-line 0
PROFILE_PMC=0 prof r06_kolmogorov64_g1c1 conv_wino4_kernel --workload kolmogorov64 --steps 4 --warmup 1 --second-line 0
ls -la gpurun_out/profiles_r06; du -sh gpurun_out
