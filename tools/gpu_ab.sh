#!/bin/bash
# parity tests + A/B of kernel variants at reduced spp + one PMC pass
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SPP=${SPP:-128}
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
for v in 0 1 0 1; do
  RT_HIP_KERNEL_VARIANT=$v timeout -k 10 300 python bench.py --spp $SPP --steps 3 --warmup 1 --cpu-tiles 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('variant $v', '%.4g rays/s' % d['value'], 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'])"
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_v1 -- python bench.py --spp $SPP --steps 1 --warmup 1 --cpu-tiles 0 > gpurun_out/pmc_v1.log 2>&1
python - <<'PY'
import csv,glob,collections
vals=collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc_v1/*/*_counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        if row['Kernel_Name']!='pt_render_tiles': continue
        vals.setdefault(row['Counter_Name'],[]).append(float(row['Counter_Value']))
for k,v in vals.items(): print(f"{k:28s}", ["%.4g"%x for x in v])
if 'SQ_THREAD_CYCLES_VALU' in vals:
    print("lane util", vals['SQ_THREAD_CYCLES_VALU'][0]/(vals['SQ_ACTIVE_INST_VALU'][0]*64))
PY
