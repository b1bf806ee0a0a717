#!/bin/bash
# kernel time + HBM traffic (FETCH_SIZE / WRITE_SIZE, one rocprofv3 --pmc pass each) of shim variants on one configuration
# usage: CONFIG=5 SPP=256 bash tools/gpu_traffic_ab.sh base c5_q128w64 ...
CONFIG=${CONFIG:-5}; SPP=${SPP:-256}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CSRC=raytracer.c_amd/csrc
ARGS="--config $CONFIG --spp $SPP --steps 2 --warmup 1 --cpu-tiles 0 --no-configs"
for name in "$@"; do
  lib=$CSRC/variants/librt_hip_$name.so; [ "$name" = base ] && lib=$CSRC/librt_hip.so
  export RT_HIP_SHIM_PATH=$PWD/$lib
  D=gpurun_out/traffic_${name}_c${CONFIG}
  rm -rf $D; mkdir -p $D
  timeout -k 10 300 python bench.py $ARGS > $D/bench.log 2>&1 || { echo "$name: bench failed"; tail -3 $D/bench.log; continue; }
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/pmc_$c -- python bench.py $ARGS > $D/pmc_$c.log 2>&1 || echo "$name: pmc $c failed"
  done
  python - "$name" "$D" <<'PY'
import csv, glob, json, sys
name, D = sys.argv[1], sys.argv[2]
b = json.loads(open(f"{D}/bench.log").read().strip().splitlines()[-1])
k = b["roofline"]["kernel"]
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{D}/pmc_{c}/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"] == k and row["Counter_Name"] == c:
                v[c] = float(row["Counter_Value"])   # the last launch
print(f"{name:22s} {k:32s} kernel_ms {b['roofline']['kernel_ms']:8.2f}  FETCH {v.get('FETCH_SIZE', 0) / 1e6:8.2f} GB(x1)  "
      f"WRITE {v.get('WRITE_SIZE', 0) / 1e6:8.2f} GB  traffic(2*F+W) {(2 * v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0)) / 1e6:8.2f} GB", flush=True)
PY
done
