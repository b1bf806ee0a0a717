#!/bin/bash
# One GPU session, parametrised (replaces round 3's nineteen one-off gpu_r3*.sh scripts; the commit log has those):
#   TESTS=1            run the -m gpu suite first and stop if it fails (default 1; TESTS=0 skips it,
#                      TESTS="tests/test_gpu_parity.py -k walls" runs a subset)
#   AB="cfg:spp[:steps[:reps]] ..."   interleaved A/Bs (tools/gpu_ab.py) of LIBS on each listed configuration
#   LIBS="base head"   shim builds to compare: base = the shipped librt_hip.so, other names = csrc/variants/librt_hip_<name>.so
#                      (make variant NAME=<name> DEFS="-D...")
#   EXTRA="cmd"        anything else, run last (its output under gpurun_out/extra.log)
# usage: gpurun --timeout 1100 -- 'AB="4:128 3:256 2:64:20 5:256:3:2" LIBS="base head" bash tools/gpu_session.sh'
TESTS=${TESTS:-1}; AB=${AB:-}; LIBS=${LIBS:-base head}; EXTRA=${EXTRA:-}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$TESTS" != "0" ]; then
  sel="tests"; [ "$TESTS" != "1" ] && sel="$TESTS"
  timeout -k 10 1000 python -m pytest $sel -m gpu -x -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
  echo "pytest exit $rc"; tail -3 gpurun_out/pytest_gpu.log
  if [ $rc -ne 0 ]; then grep -E "^(E|FAILED)" gpurun_out/pytest_gpu.log | head -30; exit 1; fi
fi
for item in $AB; do
  IFS=: read cfg spp steps reps <<< "$item"
  out=gpurun_out/ab_c${cfg}_${spp}.txt
  timeout -k 10 900 python tools/gpu_ab.py --config $cfg --spp $spp --steps ${steps:-3} --reps ${reps:-3} $LIBS > $out 2>&1 || exit 1
  tail -1 $out
done
if [ -n "$EXTRA" ]; then
  timeout -k 10 900 bash -c "$EXTRA" > gpurun_out/extra.log 2>&1; echo "extra exit $?"; tail -5 gpurun_out/extra.log
fi
