import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer.c_amd"))
import torch
from rt_amd import gpu as G, scene as S
for cfg, spp in [(4, 64), (3, 64), (5, 8)]:
    sc = S.build_scene(cfg, samples=spp)
    gs = G.GpuScene(sc)
    total = G.n_tiles(sc.width, sc.height)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t, t8, st = gs.render_tiles(1666943821, 0, 1, total, integrator="whitted")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = st.cpu().tolist()
    print(f"whitted config {cfg} {sc.width}x{sc.height}x{spp}: {dt*1e3:.1f} ms, {s[1]/dt:.3e} scans/s")
    gs.close()
