#!/usr/bin/env python3
"""Which member of the kernel family a scene takes (rt_hip_kernel_name), for a set of representative scenes, under the default
selection and under every development variant (RT_HIP_KERNEL_VARIANT exists in the development build only, librt_hip_dev.so, and is
read once per process: one child each, all on that library -- its default column is the product's pick table, pinned without a GPU
by tests/test_pick_table.py).
usage: python tools/kernel_pick_table.py            (needs a GPU: scenes are created on the device)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    for p in ("raytracer.c_amd", "tests"):
        sys.path.insert(0, os.path.join(ROOT, p))
    from rt_amd import abi, gpu as G, scene as S
    from util import packed_room, glass_scene
    scenes = []
    for c in (1, 2, 3, 4, 5):
        scenes.append((f"config {c}", S.build_scene(c, 64, 40, 2)))
    scenes.append(("glass spheres", glass_scene()))
    g5 = S.build_scene(5, 64, 40, 2, 5); g5.meshes[0].flags = abi.M_REFRACTION
    scenes.append(("config 5, glass mesh", g5))
    c5 = S.build_scene(5, 64, 40, 2); c5.objects[0].flags |= abi.M_CHECKERED
    scenes.append(("config 5, checkered wall", c5))
    g3 = S.build_scene(3, 64, 40, 2); g3.meshes[0].flags = abi.M_REFRACTION
    scenes.append(("config 3, glass cube", g3))
    scenes.append(("room of 120 spheres", packed_room(120, 1, 64, 40, 2, 5)))
    scenes.append(("room of 300 spheres", packed_room(300, 1, 64, 40, 2, 5)))
    scenes.append(("room of 300, glass", packed_room(300, 1, 64, 40, 2, 5, glass=True)))
    from util import room_with_mesh
    scenes.append(("room of 300 + 600 triangles", room_with_mesh(300, 1, 64, 40, 2, 5)))
    out = {}
    for name, sc in scenes:
        gs = G.GpuScene(sc)
        out[name] = [gs.kernel_name("path"), gs.kernel_name("whitted")]
        gs.close()
    print(json.dumps(out))
    sys.exit(0)
rows = {}
variants = ["default", "0", "2", "3", "4", "5", "7"]
for v in variants:
    env = dict(os.environ)
    env.pop("RT_HIP_KERNEL_VARIANT", None)
    env["RT_HIP_SHIM_PATH"] = os.path.join(ROOT, "raytracer.c_amd", "csrc", "librt_hip_dev.so")
    if v != "default":
        env["RT_HIP_KERNEL_VARIANT"] = v
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("variant", v, "failed:", p.stderr[-300:])
        continue
    for name, (path, whitted) in json.loads(line[-1]).items():
        rows.setdefault(name, {})[v] = (path, whitted)
for name, r in rows.items():
    d = r.get("default", ("?", "?"))
    print(f"{name:26s} trace_path: {d[0]:38s} cast_ray: {d[1]}")
    for v in variants[1:]:
        if v in r and r[v] != d:
            print(f"{'':26s}   variant {v}: {r[v][0]}" + (f" / {r[v][1]}" if r[v][1] != d[1] else ""))
