#!/bin/bash
# GPU suite on the shipped library, then three wide fuzz sweeps against the CPU oracle (random scenes, longer pools, wall rooms)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
timeout -k 10 500 python tools/gpu_fuzz_parity.py 5000 120 > gpurun_out/fuzz_a.log 2>&1; echo "fuzz a exit $?"; tail -2 gpurun_out/fuzz_a.log
FUZZ_SPP_MULT=16 timeout -k 10 500 python tools/gpu_fuzz_parity.py 6000 40 > gpurun_out/fuzz_b.log 2>&1; echo "fuzz b exit $?"; tail -2 gpurun_out/fuzz_b.log
FUZZ_WALLS=1 timeout -k 10 500 python tools/gpu_fuzz_parity.py 7000 90 > gpurun_out/fuzz_c.log 2>&1; echo "fuzz c exit $?"; tail -2 gpurun_out/fuzz_c.log
