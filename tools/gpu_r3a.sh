#!/bin/bash
# round 3, first session: the whole GPU suite (new: test_gpu_multi, test_gpu_diag) + a quick headline line
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --spp 128 --steps 5 --warmup 2 --cpu-tiles 0 --no-configs > gpurun_out/bench128.log 2>&1; tail -1 gpurun_out/bench128.log | cut -c1-300
