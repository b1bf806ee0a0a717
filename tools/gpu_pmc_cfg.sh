#!/bin/bash
# rocprofv3 --stats + PMC passes (one counter group per run, --kernel-trace only) of one configuration.
# usage: TAG=r02a CONFIG=5 SPP=64 bash tools/gpu_pmc_cfg.sh     (SPP=0: the configuration's own)
TAG=${TAG:-r02a}; CONFIG=${CONFIG:-5}; SPP=${SPP:-0}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ARGS="--config $CONFIG --spp $SPP --steps 2 --warmup 1 --cpu-tiles 0 --no-configs"
D=gpurun_out/cfg${CONFIG}_${TAG}
rm -rf $D; mkdir -p $D
timeout -k 10 600 python bench.py $ARGS > $D/bench.log 2>&1; echo "bench exit $?"; tail -1 $D/bench.log | cut -c1-200
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -- python bench.py $ARGS > $D/prof.log 2>&1; echo "rocprof exit $?"
# GRBM_GUI_ACTIVE rides in the SQ pass (GRBM has its own two slots): VALU busy is normalised by the cycles of the SAME pass
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $D/pmc_$name -- python bench.py $ARGS > $D/pmc_$name.log 2>&1; echo "pmc $name exit $?"
done
python tools/summarize_pmc_cfg.py $TAG $CONFIG
