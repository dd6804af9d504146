mkdir -p gpurun_out/r06m
timeout 900 python tools/wino4_check.py --cases 300 > gpurun_out/r06m/wino4_check.txt 2>&1
echo "w4check rc=$?"; grep -c "^ok" gpurun_out/r06m/wino4_check.txt; grep FAIL gpurun_out/r06m/wino4_check.txt | head; tail -1 gpurun_out/r06m/wino4_check.txt
timeout 1500 python -m pytest tests/test_gpu_wino4_bm64.py -x -q > gpurun_out/r06m/bm64_tests.log 2>&1
echo "bm64 rc=$?"; tail -3 gpurun_out/r06m/bm64_tests.log
timeout 900 python tests/fuzz/conv_fuzz.py --cases 300 --seed 21 > gpurun_out/r06m/conv_fuzz.txt 2>&1; tail -2 gpurun_out/r06m/conv_fuzz.txt
python tools/conv_bench.py --widths 32,64,128 --n 960 --cin 7 2>&1 | grep -v amdgpu > gpurun_out/r06m/convbench_32_64_128.txt
SDA_W4_BM64=0 python tools/conv_bench.py --widths 32,64,128 --n 960 --cin 7 2>&1 | grep -v amdgpu > gpurun_out/r06m/convbench_32_64_128_direct.txt
paste -d'|' gpurun_out/r06m/convbench_32_64_128.txt gpurun_out/r06m/convbench_32_64_128_direct.txt | cut -c1-200
