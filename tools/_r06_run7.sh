mkdir -p gpurun_out/r06g
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06g/gputest_full.log 2>&1
echo "gpu suite rc=$?"; tail -6 gpurun_out/r06g/gputest_full.log
