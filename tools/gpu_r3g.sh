#!/bin/bash
# a kernel change switched by an environment variable: suite, then interleaved A/B (same library)
# usage: SWITCH=RT_HIP_NO_BIG_PRUNE [CONFIGS="4:128 4:1024"] bash tools/gpu_r3g.sh      (SWITCH=1 is the OLD behaviour)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CONFIGS=${CONFIGS:-"4:128 4:1024"}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
for cs in $CONFIGS; do cfg=${cs%%:*}; spp=${cs##*:}; for rep in 1 2; do for m in 1 0; do
  env $SWITCH=$m timeout -k 10 300 python bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --cpu-tiles 0 --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config $cfg spp $spp $SWITCH=$m', 'kernel_ms %.3f' % d['roofline']['kernel_ms'])"
done; done; done
