#!/usr/bin/env python3
"""Interleaved A/B of shim builds on one configuration (kernel time by HIP events, bench.py).
usage: python tools/gpu_ab.py [--config 4] [--spp 128] [--reps 2] name[=path] ...
  name `base` = the shipped raytracer.c_amd/csrc/librt_hip.so; other names = csrc/variants/librt_hip_<name>.so
  (built by `make variant NAME=<name> DEFS="-D..."`)."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--spp", type=int, default=128)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
csrc = os.path.join(ROOT, "raytracer.c_amd", "csrc")
res = {}
for rep in range(a.reps):
    for spec in a.libs:
        spec, *envs = spec.split("+")
        name, _, path = spec.partition("=")
        if envs and not path and not os.path.exists(os.path.join(csrc, "variants", f"librt_hip_{name}.so")):
            path = os.path.join(csrc, "librt_hip.so")      # a switch on the shipped library
        path = path or (os.path.join(csrc, "librt_hip.so") if name == "base" else os.path.join(csrc, "variants", f"librt_hip_{name}.so"))
        env = dict(os.environ, RT_HIP_SHIM_PATH=path, **dict(e.split("=", 1) for e in envs))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(a.config), "--spp", str(a.spp),
                            "--steps", str(a.steps), "--warmup", "1", "--cpu-tiles", "0", "--no-configs"],
                           env=env, capture_output=True, text=True, timeout=600)
        try:
            d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            res.setdefault(name, []).append(d["roofline"]["kernel_ms"])
            print(f"config {a.config} spp {a.spp} {name:12s} {d['roofline']['kernel']:28s} kernel_ms {d['roofline']['kernel_ms']:.3f}  "
                  f"{d['value']:.4g} ray-bounces/s", flush=True)
        except Exception:
            print(name, "FAILED", p.stderr[-500:], flush=True)
print("summary (min ms):", {k: round(min(v), 3) for k, v in res.items()})
