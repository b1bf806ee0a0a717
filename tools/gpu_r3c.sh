#!/bin/bash
# round 3: config-5 kernel work: GPU suite, then time + traffic of variants at 4K x 256 spp and time at 64 spp
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
VARIANTS=${VARIANTS:-"r2 base"}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
CONFIG=5 SPP=256 bash tools/gpu_traffic_ab.sh $VARIANTS
timeout -k 10 600 python tools/gpu_ab.py --config 5 --spp 64 --reps 2 $VARIANTS 2>&1 | tail -3
