#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "mesh or hull or shell" > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -3 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
timeout -k 10 600 python tools/gpu_ab.py --config 5 --spp 64 --reps 2 base nodrop > gpurun_out/ab_c5_64.txt 2>&1; tail -1 gpurun_out/ab_c5_64.txt
timeout -k 10 900 python tools/gpu_ab.py --config 5 --spp 256 --reps 2 base nodrop > gpurun_out/ab_c5_256.txt 2>&1; tail -1 gpurun_out/ab_c5_256.txt
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/variants/librt_hip_phase.so timeout -k 10 300 python tools/phase.py 5 256 > gpurun_out/phase_c5_256.txt 2>&1; cat gpurun_out/phase_c5_256.txt
