#!/bin/bash
# a kernel change switched by an environment variable: suite, then interleaved A/B on the headline (same library)
# usage: SWITCH=RT_HIP_NO_BIG_PRUNE bash tools/gpu_r3g.sh      (SWITCH=1 is the OLD behaviour)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
for spp in 128 1024; do for rep in 1 2; do for m in 1 0; do
  env $SWITCH=$m timeout -k 10 300 python bench.py --spp $spp --steps 3 --warmup 1 --cpu-tiles 0 --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('spp $spp $SWITCH=$m', 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])"
done; done; done
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 4 128 > gpurun_out/diag4.log 2>&1; grep -E "VIOLATIONS|occupancy|phase-2|pruned" gpurun_out/diag4.log
