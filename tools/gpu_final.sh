#!/bin/bash
# what the driver does at round end, plus the default bench line kept under gpurun_out/
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
( time timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err ) 2>&1 | grep real; echo "bench exit $?"; tail -1 gpurun_out/bench_default.log | cut -c1-200
