mkdir -p gpurun_out/r06f
for rep in 1 2; do
for v in product s1 s2 s3; do
  if [ $v = product ]; then unset SDA_HIP_LIB; else export SDA_HIP_LIB=$PWD/sda_amd/lib_$v/libsda_hip.so; fi
  W4Q_BM64=1 python tools/w4_quick_bench.py 2>/dev/null | tail -1 >> gpurun_out/r06f/sched_variants.txt
done; done
cat gpurun_out/r06f/sched_variants.txt
