#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_gpu.log
python tools/gpu_ab.py --config 3 --spp 0 --reps 2 base tri4 2>&1 | tail -5
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 3 256 > gpurun_out/diag3b.log 2>&1; tail -8 gpurun_out/diag3b.log
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 600 python tools/diag_fuzz.py > gpurun_out/diag_fuzz.log 2>&1; echo "diag_fuzz exit $?"; tail -4 gpurun_out/diag_fuzz.log
