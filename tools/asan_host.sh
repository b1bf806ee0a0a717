#!/bin/bash
# CPU-only: the C host (raytracer.h implementation, scene builders, OBJ loader, PNG writer) and the shim's host-side hierarchy
# builder (bvh_build.h, through tests/bvh_harness.cpp) under
# AddressSanitizer + UBSan, driven by the host tests and an OBJ fuzz (GPU sanitizers are not
# available on the pool).  Leaves the regular build in place.
set -e
cd "$(dirname "$0")/.."
H=raytracer.c_amd/host
T=$(mktemp -d)
gcc -std=c99 -D_DEFAULT_SOURCE -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC \
    -Iinclude -I$H -shared -o $T/libraytracer_amd.so $H/raytracer_amd.c $H/scenes.c $H/obj_load.c $H/png_out.c \
    -Lraytracer.c_amd/csrc -lrt_hip -Wl,-rpath,$PWD/raytracer.c_amd/csrc -lz -lm
cp $H/libraytracer_amd.so $T/orig.so
trap 'cp $T/orig.so $H/libraytracer_amd.so; rm -rf $T' EXIT
cp $T/libraytracer_amd.so $H/libraytracer_amd.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
export RT_BVH_HARNESS_FLAGS="-g -fsanitize=address,undefined -fno-omit-frame-pointer"   # the hierarchy builder's checker (tests/bvh_harness.cpp)
python -m pytest tests/test_host.py tests/test_abi.py -x -q -p no:cacheprovider
RT_OBJ_FUZZ=4000 python -m pytest tests/test_host.py -x -q -p no:cacheprovider -k fuzz
