#!/usr/bin/env python3
"""Wide empirical check of the conservative rules (phase-1 filter, fp32 pre-tests, bounding-sphere probe, hull
facets): the PT_DIAG build re-tests everything they drop and counts what the exact test accepts.  Must be 0.
The driver's suite runs the reduced sets (tests/test_gpu_diag.py); this is the wider sweep, same child script.
usage: python tools/diag_fuzz.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, RT_HIP_SHIM_PATH=os.path.join(ROOT, "raytracer.c_amd", "csrc", "librt_hip_diag.so"),
           RT_HIP_DIAG_WALK_REJECTED="1")
worst = 0
for which in ("configs", "wide"):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag_child.py"), which], env=env, stdout=subprocess.PIPE, text=True)
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            r = json.loads(ln)
            if "skipped" in r:
                continue
            worst = max(worst, r["violations"])
            print(f"{r['scene']:16s} {r['integrator']:8s} {r['kernel']:32s} casts {r['casts']:10d}  candidates/cast "
                  f"{r['candidates'] / max(r['casts'], 1):6.2f}  parked {r['parked']:8d}  hull {r['left_hull_facet']:8d}  violations {r['violations']}")
    assert p.returncode == 0
assert worst == 0, "a conservative rule dropped something the exact test accepts"
print("no violations")
