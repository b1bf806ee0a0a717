#!/bin/bash
# round 3: parity + PT_DIAG checks of a kernel change, then interleaved A/B against round 2's library
# usage: VARIANTS="r2 base" bash tools/gpu_r3b.sh
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
VARIANTS=${VARIANTS:-"r2 base"}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
timeout -k 10 600 python tools/gpu_ab.py --config 4 --spp 128 --reps 2 $VARIANTS 2>&1 | tail -8
timeout -k 10 600 python tools/gpu_ab.py --config 4 --spp 1024 --reps 1 --steps 2 $VARIANTS 2>&1 | tail -4
timeout -k 10 600 python tools/gpu_ab.py --config 2 --spp 64 --reps 2 --steps 10 $VARIANTS 2>&1 | tail -6
timeout -k 10 600 python tools/gpu_ab.py --config 3 --spp 256 --reps 2 $VARIANTS 2>&1 | tail -6
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 4 128 > gpurun_out/diag4.log 2>&1; tail -12 gpurun_out/diag4.log
