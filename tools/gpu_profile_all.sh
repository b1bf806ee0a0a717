#!/bin/bash
# the round's measurement session: default bench line (as the driver runs it), smoke, then rocprofv3 --stats +
# PMC passes of configurations 4 (headline), 5 (at its own 4096 spp by default: C5SPP), 3 and 2, then the PT_DIAG
# counts (profiles/diag_c<N>.json: what bench.py's executed-work model reads).   usage: TAG=r04a bash tools/gpu_profile_all.sh
TAG=${TAG:-r04a}; C5SPP=${C5SPP:-0}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
DIAGSO=raytracer.c_amd/csrc/librt_hip_diag.so
# PT_DIAG counts first: bench.py below then finds records stamped with today's source hash
RT_HIP_SHIM_PATH=$DIAGSO timeout -k 10 300 python tools/diag.py 4 128 --json profiles/diag_c4.json > gpurun_out/diag4_$TAG.log 2>&1; echo "diag 4 exit $?"
RT_HIP_SHIM_PATH=$DIAGSO timeout -k 10 400 python tools/diag.py 5 256 --json profiles/diag_c5.json > gpurun_out/diag5_$TAG.log 2>&1; echo "diag 5 exit $?"
RT_HIP_SHIM_PATH=$DIAGSO timeout -k 10 300 python tools/diag.py 3 256 --json profiles/diag_c3.json > gpurun_out/diag3_$TAG.log 2>&1; echo "diag 3 exit $?"
RT_HIP_SHIM_PATH=$DIAGSO timeout -k 10 300 python tools/diag.py 2 64 --json profiles/diag_c2.json > gpurun_out/diag2_$TAG.log 2>&1; echo "diag 2 exit $?"
RT_HIP_SHIM_PATH=$DIAGSO timeout -k 10 300 python tools/diag.py 1 4 --json profiles/diag_c1.json > gpurun_out/diag1_$TAG.log 2>&1; echo "diag 1 exit $?"
RT_HIP_SHIM_PATH=$DIAGSO timeout -k 10 300 python tools/diag.py 5 64 > gpurun_out/diag5_64_$TAG.log 2>&1
rm -rf gpurun_out/profiles_new; mkdir -p gpurun_out/profiles_new; cp profiles/diag_c*.json gpurun_out/profiles_new/ 2>/dev/null
TAG=$TAG CONFIG=4 SPP=0 bash tools/gpu_pmc_cfg.sh 2>&1 | tail -13
TAG=$TAG CONFIG=5 SPP=$C5SPP bash tools/gpu_pmc_cfg.sh 2>&1 | tail -13
TAG=$TAG CONFIG=3 SPP=0 bash tools/gpu_pmc_cfg.sh 2>&1 | tail -13
TAG=$TAG CONFIG=2 SPP=0 bash tools/gpu_pmc_cfg.sh 2>&1 | tail -13
cp profiles/pmc_c*.json profiles/${TAG}_c* gpurun_out/profiles_new/ 2>/dev/null
# the driver's line, now with records that match the sources: parity, frac_executed, PMC keys
( time timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_full_$TAG.log 2> gpurun_out/bench_full_$TAG.err ) 2>&1 | grep real
echo "bench exit $?"; tail -1 gpurun_out/bench_full_$TAG.log | cut -c1-300
cp gpurun_out/bench_full_$TAG.log gpurun_out/profiles_new/${TAG}_bench_default.json
for c in 4 5 3 2; do cp gpurun_out/diag${c}_$TAG.log gpurun_out/profiles_new/${TAG}_diag_c${c}.txt 2>/dev/null; done
tail -4 gpurun_out/diag5_$TAG.log
