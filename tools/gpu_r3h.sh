#!/bin/bash
# GPU suite with the shipped library, then the mesh / hierarchy tests once more with a SMALL ring (128 entries, walks at 64):
# rays that find the ring full and wait a trip (`waiting`) become common
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
RT_HIP_SHIM_PATH=$PWD/raytracer.c_amd/csrc/variants/librt_hip_q128.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mesh or config5 or hull or hierarchy or walls or full_size_frames or round_mesh" > gpurun_out/pytest_q128.log 2>&1; echo "small-ring pytest exit $?"; tail -4 gpurun_out/pytest_q128.log
