#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_gpu.log
python tools/gpu_ab.py --config 3 --spp 0 --reps 2 base 2>&1 | tail -3
python tools/gpu_ab.py --config 4 --spp 128 --reps 1 base 2>&1 | tail -2
