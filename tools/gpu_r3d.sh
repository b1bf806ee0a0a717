#!/bin/bash
# config 5: variant timing at 4K x 256 spp + PT_DIAG counters of the shipped build
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python tools/gpu_ab.py --config 5 --spp 256 --reps 2 --steps 2 $VARIANTS 2>&1 | tail -3
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 5 256 > gpurun_out/diag5.log 2>&1; grep -v "^/opt" gpurun_out/diag5.log | tail -32
