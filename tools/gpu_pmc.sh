#!/bin/bash
# wave-level event counts (PT_DIAG build) + one SQ counter pass of the shipped build, config 4 at SPP
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SPP=${SPP:-128}
RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so timeout -k 10 300 python tools/diag.py 4 $SPP > gpurun_out/diag.log 2>&1; cat gpurun_out/diag.log
rm -rf gpurun_out/pmc_sq
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_sq -- python bench.py --spp $SPP --steps 1 --warmup 1 --cpu-tiles 0 > gpurun_out/pmc_sq.log 2>&1
python - <<'PY'
import csv,glob,collections
vals=collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc_sq/*/*_counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        if row['Kernel_Name']!='pt_render_tiles': continue
        vals.setdefault(row['Counter_Name'],[]).append(float(row['Counter_Value']))
for k,v in vals.items(): print(f"{k:28s}", ["%.5g"%x for x in v])
if 'SQ_THREAD_CYCLES_VALU' in vals:
    print("lane util", vals['SQ_THREAD_CYCLES_VALU'][0]/(vals['SQ_ACTIVE_INST_VALU'][0]*64))
PY
