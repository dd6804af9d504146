mkdir -p gpurun_out/r06c
timeout 1500 python -m pytest tests/test_gpu_wino4_bm64.py -x -q > gpurun_out/r06c/bm64_tests.log 2>&1
echo "bm64 rc=$?"; tail -15 gpurun_out/r06c/bm64_tests.log
timeout 900 python tests/fuzz/conv_fuzz.py --cases 400 --seed 11 > gpurun_out/r06c/conv_fuzz.txt 2>&1
echo "fuzz rc=$?"; tail -5 gpurun_out/r06c/conv_fuzz.txt
timeout 900 python tools/wino4_check.py --cases 300 > gpurun_out/r06c/wino4_check.txt 2>&1
echo "w4check rc=$?"; tail -3 gpurun_out/r06c/wino4_check.txt
