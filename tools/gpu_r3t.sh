#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -3 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
timeout -k 10 600 python tools/gpu_ab.py --config 4 --spp 128 --reps 3 base head > gpurun_out/ab_c4_128.txt 2>&1; tail -1 gpurun_out/ab_c4_128.txt
timeout -k 10 600 python tools/gpu_ab.py --config 3 --spp 256 --reps 3 base head > gpurun_out/ab_c3.txt 2>&1; tail -1 gpurun_out/ab_c3.txt
timeout -k 10 600 python tools/gpu_ab.py --config 2 --spp 64 --reps 3 --steps 20 base head > gpurun_out/ab_c2.txt 2>&1; tail -1 gpurun_out/ab_c2.txt
timeout -k 10 600 python tools/gpu_ab.py --config 5 --spp 256 --reps 2 base head > gpurun_out/ab_c5_256.txt 2>&1; tail -1 gpurun_out/ab_c5_256.txt
