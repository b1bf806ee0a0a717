#!/usr/bin/env python3
"""Empirical check of the conservative phase-1 filter: the PT_DIAG build re-tests every primitive
the filter dropped with the exact test and counts the ones the exact test accepts.  Must be 0.
usage: RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so python tools/diag_fuzz.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "raytracer.c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from rt_amd import gpu as G, scene as S
from test_gpu_parity import _random_scene
from util import whitted_scene, convex_body_scene

assert "diag" in os.environ.get("RT_HIP_SHIM_PATH", ""), "run with the PT_DIAG build"
os.environ["RT_HIP_DIAG_WALK_REJECTED"] = "1"  # also walk what bvh_probe's bounding sphere rejects, and count any triangle found
scenes = [("config %d" % c, S.build_scene(c, w, h, spp)) for c, w, h, spp in
          [(1, 256, 256, 4), (2, 400, 300, 8), (3, 240, 136, 4), (4, 480, 270, 16), (5, 96, 54, 2)]]
scenes += [("fuzz %d" % k, _random_scene(k, False, 0)) for k in range(40)]
scenes += [("fuzz mesh %d" % k, _random_scene(k, True, n)) for k, n in zip(range(100, 112), [3, 10, 40, 120, 250, 300, 400, 700, 1000, 60, 500, 2000])]
scenes += [("whitted scene", whitted_scene())]
# convex bodies at many samples per pixel: ~1e6 bounces off hull facets each, all walked (RT_HIP_DIAG_WALK_REJECTED)
scenes += [("convex body %d" % k, convex_body_scene(k, 160, 100, 64)[0]) for k in range(8)]
worst = 0
for name, sc in scenes:
    for integrator in ("path", "whitted"):
        if integrator == "whitted" and sc.max_depth > 32:
            continue
        gs = G.GpuScene(sc)
        stats = torch.zeros(48, dtype=torch.int64, device="cuda")
        try:
            gs.render_tiles(1666943821, 0, 1, G.n_tiles(sc.width, sc.height), stats=stats, integrator=integrator)
        except G.ShimError as e:
            print(f"{name:16s} {integrator:8s} skipped: {e}")
            gs.close()
            continue
        torch.cuda.synchronize()
        st = stats.cpu().tolist()
        viol, cands, casts = st[4 + 12], st[4 + 3], st[1]
        worst = max(worst, viol)
        print(f"{name:16s} {integrator:8s} casts {casts:10d}  candidates/cast {cands / max(casts, 1):6.2f}  violations {viol}")
        gs.close()
assert worst == 0, "the conservative filter dropped a primitive the exact test accepts"
print("no filter violations")
