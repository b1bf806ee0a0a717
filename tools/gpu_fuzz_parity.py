#!/usr/bin/env python3
"""One-off wide fuzz: many random scenes (tests/test_gpu_parity._random_scene), both integrators,
GPU vs the CPU oracle with the tests' parity bar.  usage: [FUZZ_SPP_MULT=24] [FUZZ_WALLS=1 | FUZZ_ROOMS=1 | FUZZ_TRIS=300,900,2500,6000] python tools/gpu_fuzz_parity.py [first] [count]
(FUZZ_TRIS: every scene gets a mesh of one of these sizes in turn -- hierarchy scenes only, for the builder)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("raytracer.c_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
import oracle_py
from rt_amd import abi, gpu as G
from test_gpu_parity import _random_scene
from util import assert_parity, fixed_point_floor

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
pt = oracle_py.PtOracle()
both = (abi.M_REFLECTION | abi.M_REFRACTION, abi.M_REFRACTION | abi.M_CHECKERED)
bad = 0
for k in range(first, first + count):
    n_tris = [0, 0, 0, 7, 60, 300, 900][k % 7]
    if os.environ.get("FUZZ_TRIS"):
        sizes = [int(v) for v in os.environ["FUZZ_TRIS"].split(",")]
        n_tris = sizes[k % len(sizes)]
    kind = ("all", "no_glass", "plain")[k % 3]  # the static (_refr), the pooled _chk / parked-walk _chk and the plain kernel families
    if os.environ.get("FUZZ_ROOMS") == "1":     # rooms of 90 .. 900 packed spheres (the streamed pooled kernels; every third keeps the generator's glass: static in-memory kernels)
        from util import packed_room
        rng_n = np.random.default_rng(k)
        sc = packed_room(int(rng_n.integers(80, 900)), k, 40 + k % 17, 24 + k % 11, 2 + k % 4, 3 + k % 6, glass=(k % 3 == 0))
    elif os.environ.get("FUZZ_WALLS") == "1":     # rooms of leading wall-sized spheres (pruned among themselves), with / without a mesh
        from util import walls_scene
        sc = walls_scene(k, width=40 + k % 17, height=24 + k % 11, samples=2 + k % 5, with_mesh=(k % 3 == 2))
    else:
        sc = _random_scene(k, n_tris > 0, n_tris, extra_flags=both if (k % 2 and kind == "all") else (), materials=kind)
    sc.samples *= int(os.environ.get("FUZZ_SPP_MULT", "1"))  # more samples per pixel: longer job pools
    for integrator in ("path", "whitted"):
        gs = G.GpuScene(sc)
        try:
            img, img8, st = gs.render_image(1666943821 + k, integrator=integrator)
        except G.ShimError as e:
            print(f"scene {k} {integrator}: refused ({str(e)[:90]})")
            gs.close()
            continue
        mean, rgb8, ost = pt.render_pixels(sc, 1666943821 + k, integrator=integrator)
        try:
            assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what=f"scene {k} {integrator}", hdr=True, abs_floor=fixed_point_floor(sc))
        except AssertionError as e:
            bad += 1
            print("FAIL", e)
        gs.close()
    if (k - first) % 50 == 49:
        print(f"{k - first + 1} scenes done, {bad} failures", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
