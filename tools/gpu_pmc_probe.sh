#!/bin/bash
# SQ counters of an arbitrary command's kernels (one rocprofv3 --pmc pass, --kernel-trace only): VALU busy and lane utilisation
# per kernel name.   usage: bash tools/gpu_pmc_probe.sh python tools/glass_mesh_probe.py 5 16
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
D=gpurun_out/pmc_probe; rm -rf $D
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $D -- "$@" > $D.log 2>&1
echo "rocprof exit $?"
python3 - $D <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
for k, c in acc.items():
    if not k.startswith("pt_render") and not k.startswith("pt_whitted"): continue
    busy = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8) if c["GRBM_GUI_ACTIVE"] else 0
    lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64) if c["SQ_ACTIVE_INST_VALU"] else 0
    print(f"{k:40s} launches {n[k]:3d}  VALU busy {100*busy:5.1f} %  lanes {100*lanes:5.1f} %  VALU insts {c['SQ_INSTS_VALU']:.3e}")
PY
