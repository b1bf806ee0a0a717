#!/usr/bin/env python3
"""Rooms of n packed spheres (beyond the LDS staging budget) with config 5's 10,240-triangle mesh in them, 1920x1080 x 16 spp:
the parked-walk body with the spheres from memory (pt_render_tiles_tri_queued_mem) against the lane-waiting pooled kernel it
replaces (RT_HIP_KERNEL_VARIANT=4 on raytracer.c_amd/csrc/librt_hip_dev.so through RT_HIP_SHIM_PATH: pt_render_tiles_pool_mem_tri).   usage: python tools/room_mesh_probe.py [n_spheres]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("raytracer.c_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import ctypes as C
import numpy as np
import torch
from rt_amd import abi, gpu as G, scene as S
from util import packed_room
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
room = packed_room(n, 3, 1920, 1080, 16, 8)
objs = [dict(flags=int(room.objects[i].flags), radius=float(room.objects[i].radius), center=room.objects[i].center.tuple(),
             color=room.objects[i].color.tuple(), emission=room.objects[i].emission.tuple()) for i in range(room.n_objects)]
c5 = S.build_scene(5)
m = c5.meshes[0]
nt = m.mesh.num_triangles
V = np.ctypeslib.as_array(C.cast(m.mesh.vertices, C.POINTER(C.c_double)), shape=(nt * 3, 5))[:, :3].reshape(nt, 3, 3)
V = V * 0.5 + np.array([0.0, -4.0, 0.0])      # into config 4's room
tris = [[tuple(t[0]), tuple(t[1]), tuple(t[2])] for t in V]
sc = S.custom_scene(objs, 1920, 1080, 16, 8, (0, 0, 50), (0, 0, 0), meshes=[dict(flags=abi.M_DEFAULT, color=(0.8, 0.7, 0.6), triangles=tris)])
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
print(f"{n + 8} spheres + {nt} triangles:", gs.kernel_name(), "%.3f ms" % best, "%.4g ray-bounces/s" % (int(st[1]) / best * 1e3))
