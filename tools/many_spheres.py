#!/usr/bin/env python3
"""What a sphere test costs on either side of the 256-sphere LDS staging budget: rooms of
8 + n spheres packed as the reference's generate_random_spheres() (main.c:65-138) would (tests/util.py packed_room),
1920x1080 x 64 spp, on the kernel the scene takes and -- through the development knobs -- on its neighbours:
  RT_HIP_FORCE_BIG=1        a small scene on the scalar-table `_big` kernel (geometry staged in LDS)
  RT_HIP_KERNEL_VARIANT=4   a scene beyond the budget on the compare-form pooled kernel (pt_render_tiles_pool_mem: no culling, no wall pruning)
  RT_HIP_KERNEL_VARIANT=3   a scene beyond the budget on round 3's static in-memory kernel (pt_render_tiles_mem)
One child process per (scene, knob): the knobs are read once per process -- and exist in the development build only
(librt_hip_dev.so, `make shim-dev`), which children with a knob load through RT_HIP_SHIM_PATH.
usage: python tools/many_spheres.py [--spp 64] [--check] 248 249 292 992 3992      (packed spheres; + 8 walls and lights)"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("raytracer.c_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
SEED = 1666943821


def child(n, spp, check, w, h, glass=False):
    import numpy as np
    import torch
    from rt_amd import gpu as G
    from util import packed_room, assert_parity, tile_pixels
    sc = packed_room(n, 1, w, h, spp, 5 if glass else 16, glass=glass)   # (glass: the reference's own MAX_DEPTH)
    gs = G.GpuScene(sc)
    total = G.n_tiles(sc.width, sc.height)
    chunks = gs.suggest_chunks(total)
    stats = torch.zeros(4, dtype=torch.int64, device="cuda")
    t, t8, _ = gs.render_tiles(SEED, 0, 1, total, chunks=chunks)          # warm
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2)]
    for a, b in ev:
        a.record()
        gs.render_tiles(SEED, 0, 1, total, t, t8, stats, chunks=chunks)
        b.record()
    torch.cuda.synchronize()
    ms = min(a.elapsed_time(b) for a, b in ev)
    casts = int(stats[1].item()) // 2
    out = {"spheres": sc.n_objects, "kernel": gs.kernel_name(), "ms": ms, "ray_bounces": casts, "chunks": chunks,
           "ns_per_sphere_test": ms * 1e6 / (casts * sc.n_objects), "ray_bounces_per_s": casts / (ms * 1e-3)}
    if check:   # 24 tiles against the CPU restatement (bit-pinned to the compiled reference): values and counters
        import oracle_py
        tiles = np.unique(np.linspace(0, total - 1, 24).astype(int))
        px = tile_pixels(sc.width, sc.height, tiles)
        mean, rgb8, ost = oracle_py.PtOracle().render_pixels(sc, SEED, pixels=px)
        rays = casts_ = 0
        vals, vals8 = [], []
        for tl in tiles:
            tt, tt8, st = gs.render_tiles(SEED, int(tl), 1, 1)
            torch.cuda.synchronize()
            vals.append(tt.cpu().numpy()[0])
            vals8.append(tt8.cpu().numpy()[0])
            rays += int(st[0].item())
            casts_ += int(st[1].item())
        assert_parity(np.concatenate(vals), np.concatenate(vals8), dict(rays=rays, tests=casts_ * sc.n_objects, casts=casts_),
                      mean, rgb8, ost, what=f"{sc.n_objects} spheres")
        out["parity"] = f"ok ({len(px)} pixels x {spp} spp against oracle/pt_oracle.c, counters equal)"
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--glass", action="store_true", help="keep the generator's M_REFRACTION spheres (a fifth), depth 5: the refraction kernels; "
                                                          "neighbour: RT_HIP_KERNEL_VARIANT=7, the static kernel")
    ap.add_argument("--child", type=int, default=-1)
    ap.add_argument("n", nargs="*", type=int)
    a = ap.parse_args()
    if a.child >= 0:
        child(a.child, a.spp, a.check, a.width, a.height, a.glass)
        sys.exit(0)
    rows = []
    for n in a.n or [248, 249, 292, 992, 3992]:
        knobs = [{}]
        if a.glass:
            knobs.append({"RT_HIP_KERNEL_VARIANT": "7"})
        elif n + 8 <= 256:
            knobs.append({"RT_HIP_FORCE_BIG": "1"})
        else:
            knobs.append({"RT_HIP_KERNEL_VARIANT": "4"})
            knobs.append({"RT_HIP_KERNEL_VARIANT": "3"})
        for kn in knobs:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", str(n), "--spp", str(a.spp), "--width", str(a.width),
                   "--height", str(a.height)] + (["--check"] if a.check and not kn else []) + (["--glass"] if a.glass else [])
            dev = {"RT_HIP_SHIM_PATH": os.path.join(ROOT, "raytracer.c_amd", "csrc", "librt_hip_dev.so")} if kn else {}
            p = subprocess.run(cmd, env=dict(os.environ, **kn, **dev), capture_output=True, text=True, timeout=1500)
            try:
                d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
                d["knob"] = " ".join(f"{k}={v}" for k, v in kn.items())
                rows.append(d)
                print(f"{d['spheres']:5d} spheres  {d['kernel']:30s} {d['knob']:26s} {d['ms']:9.2f} ms  {d['ray_bounces_per_s']:.3e} ray-bounces/s  "
                      f"{d['ns_per_sphere_test'] * 1e3:7.3f} ps per sphere test  {d.get('parity', '')}", flush=True)
            except Exception:
                print(n, kn, "FAILED", p.stderr[-800:], flush=True)
    print(json.dumps(rows))
