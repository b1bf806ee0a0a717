#!/bin/bash
# A/B of alternative shim builds on one configuration: CFG=3 SPP=0 bash tools/gpu_ab_cfg.sh lib1.so lib2.so ...
CFG=${CFG:-3}; SPP=${SPP:-0}
for rep in 1 2; do
for lib in "$@"; do
  RT_HIP_SHIM_PATH=$lib timeout -k 10 300 python bench.py --config $CFG --spp $SPP --steps 3 --warmup 1 --cpu-tiles 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config $CFG', '$lib', '%.4g rays/s' % d['value'], 'ms %.3f' % d['ms_per_step'])"
done; done
