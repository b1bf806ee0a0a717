#!/usr/bin/env python3
"""What config 5 would cost if its triangles were free: the same room with a true SPHERE (radius 8, same centre, same
material) in place of the 10,240-triangle tessellated one, on the pooled sphere kernel -- against the real thing on the
parked-walk kernel.  Paths are nearly the same (the mesh is that sphere to within its facets).  usage: python tools/c5_floor.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "raytracer.c_amd"))
import torch
from rt_amd import abi, gpu as G, scene as S
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
real = S.build_scene(5, samples=spp)
objs = [dict(flags=real.objects[i].flags, radius=real.objects[i].radius, center=real.objects[i].center.tuple(),
             color=real.objects[i].color.tuple(), emission=real.objects[i].emission.tuple()) for i in range(real.n_objects)]
m = real.meshes[0]
objs.append(dict(flags=m.flags, radius=8.0, center=(0.0, -8.0, 4.0), color=m.color.tuple(), emission=m.emission.tuple()))
info = S.scene_info(5)
ball = S.custom_scene(objs, real.width, real.height, spp, real.max_depth, tuple(info.cam_pos), tuple(info.cam_target))
# ... and the true-sphere room with the tessellated sphere moved OUTSIDE the room (200 units up, beyond the ceiling wall): the
# parked-walk kernel with its probe for every ray, but nothing ever parked or walked
import math
def uv_sphere(radius, c, lon_n=80, lat_n=64):
    def v(lon, lat):
        phi, theta = 2 * math.pi * lon / lon_n, math.pi * lat / lat_n
        return (c[0] + radius * math.sin(theta) * math.cos(phi), c[1] + radius * math.cos(theta), c[2] + radius * math.sin(theta) * math.sin(phi))
    tris = []
    for lat in range(lat_n):
        for lon in range(lon_n):
            a, b, cc, d = v(lon, lat), v(lon + 1, lat), v(lon + 1, lat + 1), v(lon, lat + 1)
            tris += [(a, cc, b), (a, d, cc)]
    return tris
far = S.custom_scene(objs, real.width, real.height, spp, real.max_depth, tuple(info.cam_pos), tuple(info.cam_target),
                     meshes=[dict(flags=m.flags, color=m.color.tuple(), emission=(0, 0, 0), triangles=uv_sphere(8.0, (0.0, 200.0, 4.0)))])
for name, sc in (("config 5 (10,240 triangles)", real), ("the same room, a true sphere", ball), ("true sphere + mesh out of reach", far)):
    gs = G.GpuScene(sc)
    n = G.n_tiles(sc.width, sc.height)
    stats = torch.zeros(abi.NSTATS, dtype=torch.int64, device="cuda")
    tiles = tiles8 = None
    tiles, tiles8, _ = gs.render_tiles(1666943821, 0, 1, n, stats=stats)
    torch.cuda.synchronize()
    stats.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        gs.render_tiles(1666943821, 0, 1, n, tiles, tiles8, stats)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    casts = stats.cpu().tolist()[1] / 3
    print(f"{name:32s} {gs.kernel_name():32s} {ms:8.2f} ms  {casts:.4g} ray-bounces  {casts / ms / 1e6:.2f} G ray-bounces/s", flush=True)
