#!/bin/bash
mkdir -p gpurun_out/r2k
W4_TRACE_DEBUG=4096,0 timeout 900 python tools/wino4_check.py --skip-check --variants 11 > gpurun_out/r2k/trace3.txt 2>&1
grep "===\|wave [04]\|trace" gpurun_out/r2k/trace3.txt
