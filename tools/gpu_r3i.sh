#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
timeout -k 10 600 python tools/gpu_ab.py --config 2 --spp 64 --reps 3 --steps 20 $VARIANTS 2>&1 | tail -1
timeout -k 10 600 python tools/gpu_ab.py --config 3 --spp 256 --reps 2 $VARIANTS 2>&1 | tail -1
timeout -k 10 600 python tools/gpu_ab.py --config 4 --spp 128 --reps 2 $VARIANTS 2>&1 | tail -1
timeout -k 10 600 python tools/gpu_ab.py --config 4 --spp 1024 --reps 1 --steps 3 $VARIANTS 2>&1 | tail -1
timeout -k 10 600 python tools/gpu_ab.py --config 1 --spp 4 --reps 2 --steps 50 $VARIANTS 2>&1 | tail -1
