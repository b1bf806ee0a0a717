#!/bin/bash
# how a configuration's kernel responds to fewer resident workgroups per CU (RT_HIP_EXTRA_LDS: the development build, librt_hip_dev.so)
CONFIG=${CONFIG:-5}; SPP=${SPP:-64}
for x in 0 9000 21000 48000; do
  RT_HIP_SHIM_PATH=$PWD/raytracer.c_amd/csrc/librt_hip_dev.so RT_HIP_EXTRA_LDS=$x timeout -k 10 300 python bench.py --config $CONFIG --spp $SPP --steps 3 --warmup 1 --cpu-tiles 0 --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('extra LDS $x:', '%.2f ms' % d['roofline']['kernel_ms'], d['roofline']['kernel'])"
done
