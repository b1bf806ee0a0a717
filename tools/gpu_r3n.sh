#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 5 64 > gpurun_out/diag_c5_64.txt 2>&1; echo "diag exit $?"; grep -E "VIOLATIONS|node-visit|parked rays per|dropped" gpurun_out/diag_c5_64.txt
RT_HIP_DIAG_WALK_REJECTED=1 RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 5 16 > gpurun_out/diag_c5_16_wr.txt 2>&1; echo "diag exit $?"; grep -E "VIOLATIONS|node-visit|parked rays per|dropped" gpurun_out/diag_c5_16_wr.txt
