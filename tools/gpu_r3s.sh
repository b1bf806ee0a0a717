#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -3 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
timeout -k 10 600 python tools/gpu_ab.py --config 5 --spp 64 --reps 2 base head > gpurun_out/ab_c5_64.txt 2>&1; tail -1 gpurun_out/ab_c5_64.txt
timeout -k 10 600 python tools/gpu_ab.py --config 5 --spp 256 --reps 2 base head > gpurun_out/ab_c5_256.txt 2>&1; tail -1 gpurun_out/ab_c5_256.txt
RT_HIP_DIAG_WALK_REJECTED=1 RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 5 16 > gpurun_out/diag_c5_16_wr.txt 2>&1; echo "diag exit $?"; grep -E "VIOLATIONS|cannot see|hull facet" gpurun_out/diag_c5_16_wr.txt
