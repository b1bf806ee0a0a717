#!/bin/bash
# round 3, session m: distance-coded traversal stacks (waiting boxes dropped at the pop) -- suite, PT_DIAG counts, A/B against HEAD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -3 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 5 64 > gpurun_out/diag_c5_64.txt 2>&1; echo "diag exit $?"; grep -E "VIOLATIONS|node-visit|parked rays per" gpurun_out/diag_c5_64.txt; tail -12 gpurun_out/diag_c5_64.txt | head -3
timeout -k 10 600 python tools/gpu_ab.py --config 5 --spp 64 --reps 2 base head > gpurun_out/ab_c5_64.txt 2>&1; tail -5 gpurun_out/ab_c5_64.txt
timeout -k 10 900 python tools/gpu_ab.py --config 5 --spp 256 --reps 2 base head > gpurun_out/ab_c5_256.txt 2>&1; tail -5 gpurun_out/ab_c5_256.txt
