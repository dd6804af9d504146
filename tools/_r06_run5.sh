mkdir -p gpurun_out/r06e
export SDA_HIP_LIB=$PWD/sda_amd/lib_trace/libsda_hip.so
W4_TRACE_BM64=1 W4_TRACE_DEBUG=8192 timeout 600 python tools/wino4_check.py --variants 11 --skip-check > gpurun_out/r06e/trace_mf2.txt 2>&1
W4_TRACE_DEBUG=8192 W4_TRACE_CASES="96->96 @64,384->384" timeout 600 python tools/wino4_check.py --variants 11 --skip-check > gpurun_out/r06e/trace_mf3.txt 2>&1
W4_TRACE_BM64=1 W4_TRACE_DEBUG=0 W4_TRACE_CASES="64->64 @64,256->256" timeout 600 python tools/wino4_check.py --variants 11 --skip-check > gpurun_out/r06e/trace_mf2_stamps.txt 2>&1
grep -v "^$" gpurun_out/r06e/trace_mf2.txt | grep -v "wave [1-3567]:"
grep -v "^$" gpurun_out/r06e/trace_mf3.txt | grep -v "wave [1-3567]:"
grep -v "^$" gpurun_out/r06e/trace_mf2_stamps.txt | grep -v "wave [1-3567]:"
