#!/usr/bin/env python3
"""Interleaved A/B of shim builds on bench.py's two `integrators` workloads (the glass scene on trace_path's M_REFRACTION
kernels, cast_ray on config 4's room), kernel time by HIP events.  usage: python tools/integrators_ab.py [--reps 3] base r3 ..."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
csrc = os.path.join(ROOT, "raytracer.c_amd", "csrc")
res = {}
for rep in range(a.reps):
    for name in a.libs:
        path = os.path.join(csrc, "librt_hip.so") if name == "base" else os.path.join(csrc, "variants", f"librt_hip_{name}.so")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--integrators-only"], env=dict(os.environ, RT_HIP_SHIM_PATH=path),
                           capture_output=True, text=True, timeout=600)
        try:
            d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            for e in d["integrators"]:
                res.setdefault((name, e["kernel"]), []).append(e["kernel_ms"])
                print(f"{name:8s} {e['kernel']:26s} {e['kernel_ms']:9.3f} ms  {e['ray_bounces_per_s']:.4g} scene scans/s", flush=True)
        except Exception:
            print(name, "FAILED", p.stderr[-500:], flush=True)
print("summary (min ms):", {f"{k[0]}:{k[1]}": round(min(v), 3) for k, v in res.items()})
