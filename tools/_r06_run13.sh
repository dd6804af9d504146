mkdir -p gpurun_out/r06o
timeout 1700 python tests/fuzz/net_fuzz.py --cases 250 --seed 61 > gpurun_out/r06o/net_fuzz.txt 2>&1; echo "net rc=$?"; tail -3 gpurun_out/r06o/net_fuzz.txt
timeout 1200 python tests/fuzz/path_fuzz.py --cases 150 --seed 62 > gpurun_out/r06o/path_fuzz.txt 2>&1; echo "path rc=$?"; tail -3 gpurun_out/r06o/path_fuzz.txt
