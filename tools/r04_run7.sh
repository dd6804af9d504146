#!/bin/bash
mkdir -p gpurun_out/r4g
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/bf16x6_probe.hip -o /tmp/bf16x6_probe && /tmp/bf16x6_probe > gpurun_out/r4g/bf16x6_probe.txt 2>&1; cat gpurun_out/r4g/bf16x6_probe.txt
SDA_HIP_LIB=sda_amd/lib_w4v/libsda_hip.so python tools/bf16x6_pipeline_probe.py 2>&1 | grep -v Warn | tee gpurun_out/r4g/bf16x6_pipeline.txt
