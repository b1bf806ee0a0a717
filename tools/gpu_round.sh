#!/bin/bash
# one GPU session: tests, smoke, peak microbench, headline bench, rocprof summary
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
timeout -k 10 120 ./tools/fp64_peak > gpurun_out/fp64_peak.log 2>&1; cat gpurun_out/fp64_peak.log
timeout -k 10 600 python bench.py > gpurun_out/bench_full.log 2>&1; echo "bench exit $?"; tail -2 gpurun_out/bench_full.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python bench.py --steps 2 --warmup 1 --cpu-tiles 0 > gpurun_out/prof_bench.log 2>&1; echo "rocprof exit $?"; tail -2 gpurun_out/prof_bench.log
find gpurun_out/prof_r01 -name "*stats*" | head
