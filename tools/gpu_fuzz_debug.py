#!/usr/bin/env python3
"""Details on single fuzz scenes: python tools/gpu_fuzz_debug.py 1081 1124 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("raytracer.c_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
import torch
import oracle_py
from rt_amd import abi, gpu as G
from test_gpu_parity import _random_scene

pt = oracle_py.PtOracle()
both = (abi.M_REFLECTION | abi.M_REFRACTION, abi.M_REFRACTION | abi.M_CHECKERED)
for k in [int(a) for a in sys.argv[1:]]:
    n_tris = [0, 0, 0, 7, 60, 300, 900][k % 7]
    sc = _random_scene(k, n_tris > 0, n_tris, extra_flags=both if k % 2 else ())
    flags = sorted({int(sc.objects[i].flags) for i in range(sc.n_objects)})
    emax = max([max(sc.objects[i].emission.x, sc.objects[i].emission.y, sc.objects[i].emission.z) for i in range(sc.n_objects)] + [0])
    rmin = min(abs(sc.objects[i].radius) for i in range(sc.n_objects))
    rmax = max(abs(sc.objects[i].radius) for i in range(sc.n_objects))
    print(f"scene {k}: {sc.width}x{sc.height} spp {sc.samples} depth {sc.max_depth} spheres {sc.n_objects} tris {sc.n_triangles} "
          f"flags {flags} max emission {emax:.3g} radii {rmin:.3g}..{rmax:.3g}")
    gs = G.GpuScene(sc)
    seed = 1666943821 + k
    stats = torch.zeros(48, dtype=torch.int64, device="cuda")
    total = G.n_tiles(sc.width, sc.height)
    t, t8, _ = gs.render_tiles(seed, 0, 1, total, stats=stats)
    img, img8 = gs.untile(t, t8, 0, 1, total)
    torch.cuda.synchronize()
    st = stats.cpu().tolist()
    mean, rgb8, ost = pt.render_pixels(sc, seed)
    g = img.cpu().numpy().reshape(-1, 3).astype(np.float64)
    err = np.abs(g - mean)
    bad = np.argwhere(err > 1e-6 * np.abs(mean) + 1e-12)
    print(f"  gpu rays {st[0]} casts {st[1]} tests {st[2]} | oracle rays {ost['rays']} casts {ost['casts']} tests {ost['tests']}"
          f" | diag violations {st[4 + 12]}")
    print(f"  {len(bad)} values off; worst abs {err.max():.3g} at pixel {np.unravel_index(err.argmax(), err.shape)} "
          f"gpu {g.reshape(-1)[err.argmax()]:.6g} oracle {mean.reshape(-1)[err.argmax()]:.6g}; pixels off: {sorted(set(bad[:, 0].tolist()))[:12]}")
    gs.close()
