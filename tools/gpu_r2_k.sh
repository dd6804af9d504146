#!/bin/bash
mkdir -p gpurun_out/r2k
W4_TRACE_DEBUG=4096 timeout 900 python tools/wino4_check.py --cases 60 --variants 11 > gpurun_out/r2k/check2.txt 2>&1; echo "rc=$?" >> gpurun_out/r2k/check2.txt
grep -v "^ok\|^   wave [123567]" gpurun_out/r2k/check2.txt | tail -50
