#!/bin/bash
for spp in 64 512; do
  RT_HIP_KERNEL_VARIANT=2 python tools/gpu_ab.py --config 5 --spp $spp --reps 1 --steps 2 base 2>&1 | grep "^config"
  python tools/gpu_ab.py --config 5 --spp $spp --reps 1 --steps 2 base 2>&1 | grep "^config"
done
