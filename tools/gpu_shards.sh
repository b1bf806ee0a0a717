#!/bin/bash
# one rank's share of the headline frame rendered alone on one GPU (bench.py --shard R/N): what rt_hip_suggest_chunks picks, and the chunk counts around it
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
show() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', '%.3f ms/step' % d['ms_per_step'], 'chunks', d['config']['sample_chunks_per_tile'])"; }
for n in 1 2 4 8; do
  timeout -k 10 300 python bench.py --shard 0/$n --steps 5 --warmup 2 --cpu-tiles 0 --no-configs 2>/dev/null | show "shard 0/$n auto"
done
for k in 5 6 7 8 9 10; do
  timeout -k 10 300 python bench.py --shard 0/8 --chunks $k --steps 5 --warmup 2 --cpu-tiles 0 --no-configs 2>/dev/null | show "shard 0/8 chunks $k"
done
for k in 2 3 4 5; do
  timeout -k 10 300 python bench.py --shard 0/4 --chunks $k --steps 5 --warmup 2 --cpu-tiles 0 --no-configs 2>/dev/null | show "shard 0/4 chunks $k"
done
for k in 3 5; do
  timeout -k 10 300 python bench.py --shard 3/8 --chunks 7 --steps 5 --warmup 2 --cpu-tiles 0 --no-configs 2>/dev/null | show "shard 3/8 chunks 7"
done
