#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mesh or config5 or hierarchy or whitted_configs or fuzz_mesh or two_cameras or checkered" > gpurun_out/pytest_mesh.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/pytest_mesh.log
python tools/gpu_ab.py --config 5 --spp 64 --reps 1 base q3 2>&1 | tail -4
RT_HIP_KERNEL_VARIANT=2 python tools/gpu_ab.py --config 5 --spp 64 --reps 1 base 2>&1 | tail -2
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 5 64 > gpurun_out/diag5b.log 2>&1; tail -12 gpurun_out/diag5b.log
