#!/bin/bash
# PMC passes for pt_render_tiles (separate runs: counters only with --kernel-trace)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SPP=${SPP:-128}
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -c "" gpurun_out/counters_list.txt
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- python bench.py --spp $SPP --steps 1 --warmup 1 --cpu-tiles 0 > gpurun_out/pmc_$name.log 2>&1
  echo "pmc $name exit $?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
for d in gpurun_out/pmc_*/; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $f"; [ -n "$f" ] && grep pt_render_tiles "$f" | awk -F, '{print $(NF-1), $NF}' | sort | uniq -c | head -20; done
