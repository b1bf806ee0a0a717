#!/bin/bash
# per-kernel register / scratch / LDS / instruction statistics of the device code (gfx950)
# usage: tools/isa_stats.sh [extra hipcc flags]   -> prints one line per kernel
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${ISA_OUT:-/tmp/isa/pt_kernel.s}
mkdir -p "$(dirname "$OUT")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$ROOT/include -I$ROOT/raytracer.c_amd/csrc \
    --cuda-device-only -S -o "$OUT" "$@" $ROOT/raytracer.c_amd/csrc/pt_kernel.hip || exit 1
python3 - "$OUT" "$ROOT" "${ISA_JSON:-}" <<'PY'
import json, os, re, sys
txt = open(sys.argv[1]).read()
root, json_out = sys.argv[2], sys.argv[3]
recs = {}
# kernel bodies: from "name:" to ".end_amdhsa_kernel" metadata; use .amdhsa blocks for resources
for m in re.finditer(r"\.amdhsa_kernel (\w+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, blk = m.group(1), m.group(2)
    def g(key):
        r = re.search(r"\.amdhsa_%s (\d+)" % key, blk)
        return int(r.group(1)) if r else -1
    body = re.search(r"^%s:[^\n]*\n(.*?)^\.Lfunc_end\d+:" % re.escape(name), txt, re.S | re.M)  # the whole function (a kernel may hold several s_endpgm)
    b = body.group(1) if body else ""
    cnt = lambda pat: len(re.findall(pat, b, re.M))
    n_all, n_f64 = cnt(r"^\s+[vsd][_a-z]"), cnt(r"^\s+v_\w+_f64")
    n_pk, n_scr = cnt(r"^\s+v_pk_\w+_f32"), cnt(r"^\s+scratch_")
    n_lane = cnt(r"^\s+v_(read|write)lane_b32")  # mostly SGPRs spilled to VGPR lanes: a VALU slot each
    recs[name] = {"vgpr": g('next_free_vgpr'), "sgpr": g('next_free_sgpr'), "scratch_bytes": g('private_segment_fixed_size'),
                  "lds_static_bytes": g('group_segment_fixed_size'), "scratch_ops": n_scr, "instructions": n_all, "lane_ops": n_lane}
    print(f"{name:34s} vgpr {g('next_free_vgpr'):4d} sgpr {g('next_free_sgpr'):4d} scratch {g('private_segment_fixed_size'):5d} "
          f"lds {g('group_segment_fixed_size'):6d} | insts {n_all:6d} f64 {n_f64:5d} pk_f32 {n_pk:4d} scratch_ops {n_scr:4d} lane_ops {n_lane:4d}")
if json_out:   # ISA_JSON=profiles/isa_stats.json tools/isa_stats.sh : the record bench.py reports registers / scratch from
    sys.path.insert(0, root)
    from bench import kernel_source_sha256
    json.dump({"source_sha256": kernel_source_sha256(), "flags": "hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off (the shim's own)",
               "kernels": recs}, open(json_out, "w"), indent=1)
    print("wrote", json_out)
PY
