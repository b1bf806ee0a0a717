#!/bin/bash
# rocprofv3 --stats + PMC passes (one counter group per run, --kernel-trace only) of one configuration.
# usage: TAG=r04a CONFIG=5 SPP=64 bash tools/gpu_pmc_cfg.sh     (SPP=0: the configuration's own)
TAG=${TAG:-r04a}; CONFIG=${CONFIG:-5}; SPP=${SPP:-0}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# the un-profiled timing leg measures the kernel the way bench.py's driver line does for this configuration (the PMC record's
# kernel_ms is compared with it, 3 %): the headline 20 frames after 5 warm ones; configs 1, 2 the same; config 3 three
# frames after one; config 5 one after one
case $CONFIG in
  4|1|2) TIMING="--steps 20 --warmup 5";;
  3)     TIMING="--steps 3 --warmup 1";;
  *)     TIMING="--steps 1 --warmup 1";;
esac
COMMON="--config $CONFIG --spp $SPP --cpu-tiles 0 --no-configs"
ARGS="$COMMON --steps 2 --warmup 1"
D=gpurun_out/cfg${CONFIG}_${TAG}
rm -rf $D; mkdir -p $D
timeout -k 10 600 python bench.py $COMMON $TIMING > $D/bench.log 2>&1; echo "bench exit $?"; tail -1 $D/bench.log | cut -c1-200
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -- python bench.py $ARGS > $D/prof.log 2>&1; echo "rocprof exit $?"
# GRBM_GUI_ACTIVE rides in the SQ pass (GRBM has its own two slots): VALU busy is normalised by the cycles of the SAME pass
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $D/pmc_$name -- python bench.py $ARGS > $D/pmc_$name.log 2>&1; echo "pmc $name exit $?"
done
python tools/summarize_pmc_cfg.py $TAG $CONFIG
