#!/bin/bash
mkdir -p gpurun_out/r2c
timeout 900 python tools/wino4_check.py --cases 30 --variants 0,1,2,3,4,5,6 --bench > gpurun_out/r2c/wino4_check.txt 2>&1; echo "rc=$?" >> gpurun_out/r2c/wino4_check.txt
SDA_CONV_DEBUG=8 timeout 300 python tools/wino4_check.py --skip-check --variants 0,6 > gpurun_out/r2c/noepi.txt 2>&1
grep -v "^ok" gpurun_out/r2c/wino4_check.txt | tail -40; echo ---- no epilogue stores; cat gpurun_out/r2c/noepi.txt
