#!/usr/bin/env python3
"""config 3's scene (the cube mesh + 5 spheres) with one sphere and the cube turned M_REFRACTION, 1920x1080 x 64 spp, depth 8:
the pooled refraction kernel for small mesh scenes (pt_render_tiles_tri_refr_pool) against the static one (RT_HIP_KERNEL_VARIANT=7, with RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_dev.so: the switch exists in the development build only).
`python tools/glass_mesh_probe.py 5 [spp]`: config 5's scene instead (10,240 triangles through the hierarchy, 3840x2160, depth 5 as the
reference's MAX_DEPTH): first as it is (the parked-walk kernel), then with the mesh turned M_REFRACTION -- the family that still
runs on the static body with lane-waiting walks (pt_render_tiles_tri_big_refr)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "raytracer.c_amd"))
import torch
from rt_amd import abi, gpu as G, scene as S
def run(sc, what):
    gs = G.GpuScene(sc)
    total = G.n_tiles(sc.width, sc.height)
    st = torch.zeros(4, dtype=torch.int64, device="cuda")
    t, t8, _ = gs.render_tiles(1666943821, 0, 1, total)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.zero_(); a.record(); gs.render_tiles(1666943821, 0, 1, total, t, t8, st); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    print(what, gs.kernel_name(), "%.3f ms" % best, "%.4g scene scans/s" % (int(st[1]) / best * 1e3), flush=True)
    gs.close()


if len(sys.argv) > 1 and sys.argv[1] == "check":
    # small glass-mesh hierarchy scenes against the oracle (the tests' bar), both kernels of the family
    for p_ in ("oracle", "tests"):
        sys.path.insert(0, os.path.join(ROOT, p_))
    import numpy as np
    import oracle_py
    from util import assert_parity, fixed_point_floor
    pt = oracle_py.PtOracle()
    for (w, h, spp, depth, glass_sphere) in ((64, 36, 3, 5, False), (96, 54, 6, 8, True), (40, 24, 24, 3, True)):
        sc = S.build_scene(5, w, h, spp, depth)
        sc.meshes[0].flags = abi.M_REFRACTION
        if glass_sphere:
            sc.objects[sc.n_objects - 1].flags = abi.M_REFRACTION
        mean, rgb8, ost = pt.render_pixels(sc, 1666943821)
        gs = G.GpuScene(sc)   # (RT_HIP_KERNEL_VARIANT=7 in the environment: the static kernel of the family)
        img, img8, st = gs.render_image(1666943821)
        assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what=gs.kernel_name(), hdr=True, abs_floor=fixed_point_floor(sc))
        print(w, h, spp, depth, gs.kernel_name(), "ok", st, flush=True)
        gs.close()
elif len(sys.argv) > 1 and sys.argv[1] == "5":
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    sc = S.build_scene(5, None, None, spp, 5)
    run(sc, f"config 5 x {spp} spp, depth 5:")
    sc.meshes[0].flags = abi.M_REFRACTION
    run(sc, "  the mesh M_REFRACTION:")
else:
    sc = S.build_scene(3, None, None, 64)
    sc.objects[1].flags = abi.M_REFRACTION
    sc.meshes[0].flags = abi.M_REFRACTION
    run(sc, "config 3, one sphere + the cube M_REFRACTION:")
