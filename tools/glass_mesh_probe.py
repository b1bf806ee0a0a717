#!/usr/bin/env python3
"""config 3's scene (the cube mesh + 5 spheres) with one sphere and the cube turned M_REFRACTION, 1920x1080 x 64 spp, depth 8:
the pooled refraction kernel for small mesh scenes (pt_render_tiles_tri_refr_pool) against the static one (RT_HIP_KERNEL_VARIANT=7)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "raytracer.c_amd"))
import torch
from rt_amd import abi, gpu as G, scene as S
sc = S.build_scene(3, None, None, 64)
sc.objects[1].flags = abi.M_REFRACTION
sc.meshes[0].flags = abi.M_REFRACTION
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
print(gs.kernel_name(), "%.3f ms" % best, "%.4g scene scans/s" % (int(st[1]) / best * 1e3))
