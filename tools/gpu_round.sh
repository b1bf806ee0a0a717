#!/bin/bash
# one GPU session: tests, smoke, peak microbench, headline bench, rocprof summary, PMC passes
TAG=${TAG:-r01e}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench_full.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_full.log | cut -c1-260
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_${TAG}_*
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python bench.py --steps 2 --warmup 1 --cpu-tiles 0 > gpurun_out/prof_bench.log 2>&1; echo "rocprof exit $?"
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python bench.py --steps 1 --warmup 1 --cpu-tiles 0 > /dev/null 2>&1; echo "pmc $name exit $?"
done
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec head -3 {} \;
