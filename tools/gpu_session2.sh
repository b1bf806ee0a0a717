#!/bin/bash
bash tools/gpu_quick.sh
python tools/gpu_ab.py --config 3 --spp 0 base tri4 2>&1 | tail -5
python tools/gpu_ab.py --config 2 --spp 0 base 2>&1 | tail -2
for v in "" wh3 wh2; do
  if [ -z "$v" ]; then lib=raytracer.c_amd/csrc/librt_hip.so; else lib=raytracer.c_amd/csrc/variants/librt_hip_$v.so; fi
  echo "whitted variant: ${v:-base}"; RT_HIP_SHIM_PATH=$lib timeout -k 10 300 python tools/whitted_bench.py 2>&1 | grep whitted
done
