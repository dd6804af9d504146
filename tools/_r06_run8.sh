mkdir -p gpurun_out/r06h
timeout 1500 python -m pytest tests/test_gpu_kolmogorov_eval.py -x -q -s -k "32_step_chain" > gpurun_out/r06h/chain32.log 2>&1
echo "chain rc=$?"; grep -v amdgpu gpurun_out/r06h/chain32.log | tail -8
timeout 900 python tools/h2_trained_like.py > gpurun_out/r06h/h2_trained_like.txt 2>&1
grep -v amdgpu gpurun_out/r06h/h2_trained_like.txt
