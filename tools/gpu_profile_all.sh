#!/bin/bash
# the round's measurement session: default bench line (as the driver runs it), smoke, then rocprofv3 --stats +
# PMC passes of configurations 4 (headline), 5 (at 256 spp), 3 and 2.   usage: TAG=r02d bash tools/gpu_profile_all.sh
TAG=${TAG:-r03c}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_full_$TAG.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_full_$TAG.log | cut -c1-300
TAG=$TAG CONFIG=4 SPP=0 bash tools/gpu_pmc_cfg.sh 2>&1 | tail -13
TAG=$TAG CONFIG=5 SPP=256 bash tools/gpu_pmc_cfg.sh 2>&1 | tail -13
TAG=$TAG CONFIG=3 SPP=0 bash tools/gpu_pmc_cfg.sh 2>&1 | tail -13
TAG=$TAG CONFIG=2 SPP=0 bash tools/gpu_pmc_cfg.sh 2>&1 | tail -13
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 4 128 > gpurun_out/diag4_$TAG.log 2>&1
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 5 256 > gpurun_out/diag5_$TAG.log 2>&1
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 5 64 > gpurun_out/diag5_64_$TAG.log 2>&1
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 3 256 > gpurun_out/diag3_$TAG.log 2>&1
tail -6 gpurun_out/diag5_$TAG.log
