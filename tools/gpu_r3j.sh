#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
for cs in $CONFIGS; do cfg=${cs%%:*}; spp=${cs##*:}; timeout -k 10 600 python tools/gpu_ab.py --config $cfg --spp $spp --reps 2 --steps 3 $VARIANTS 2>&1 | tail -1; done
