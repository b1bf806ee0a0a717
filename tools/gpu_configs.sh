#!/bin/bash
# the other BASELINE configurations and the per-rank shares of the headline frame, one GPU
mkdir -p gpurun_out
show() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', '%.4g ray-bounces/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], '%.0f Mpixel-samples/s' % d['mpixel_samples_per_s'], 'frac %.3f' % d['roofline']['frac'])"; }
for c in 1 2 3; do
  timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 2 --cpu-tiles 0 2>/dev/null | show "config $c"
done
timeout -k 10 600 python bench.py --config 5 --spp 64 --steps 2 --warmup 1 --cpu-tiles 0 2>/dev/null | show "config 5 @64spp"
for n in 1 2 4 8; do
  timeout -k 10 300 python bench.py --shard 0/$n --steps 3 --warmup 1 --cpu-tiles 0 2>/dev/null | show "shard 0/$n"
done
