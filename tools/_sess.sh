#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "mesh or config5 or hierarchy or soup or golden or tri or fuzz" > gpurun_out/pytest_leaf.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -2 gpurun_out/pytest_leaf.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/gpu_ab.py --config 5 --spp 64 --reps 2 head base
timeout -k 10 600 python tools/gpu_ab.py --config 5 --spp 256 --reps 1 head base
