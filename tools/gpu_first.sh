#!/bin/bash
# first GPU contact: parity tests + a short timing of config 4
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --spp 64 --steps 2 --warmup 1 --cpu-tiles 0 > gpurun_out/bench_spp64.log 2>&1; echo "bench exit $?"
tail -5 gpurun_out/bench_spp64.log
