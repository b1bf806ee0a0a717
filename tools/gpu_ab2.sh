#!/bin/bash
# A/B of alternative shim builds (RT_HIP_SHIM_PATH), interleaved
SPP=${SPP:-128}
for rep in 1 2; do
for lib in "$@"; do
  RT_HIP_SHIM_PATH=$lib timeout -k 10 300 python bench.py --spp $SPP --steps 3 --warmup 1 --cpu-tiles 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', '%.4g rays/s' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'])"
done; done
