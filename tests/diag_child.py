#!/usr/bin/env python3
"""Child process of tests/test_gpu_diag.py (and tools/diag_fuzz.py): renders scenes with the PT_DIAG build of the
shim (RT_HIP_SHIM_PATH=.../librt_hip_diag.so, RT_HIP_DIAG_WALK_REJECTED=1) and prints one JSON line per
(scene, integrator) with the counters the conservative rules are judged by:

  violations   stats[4 + 12]: primitives the packed-fp32 filter or an fp32 triangle pre-test dropped although the
               exact fp64 test accepts them, PLUS rays the bounding-sphere probe or the hull-facet rule would not
               have walked that, walked all the same, came back with a triangle.  Must be 0.
               PLUS wall-sized spheres pruned before the exact tests (BigPrune) that the exact test, run all the same, finds
               no farther than the scan's result.
  and what shows that the check was not vacuous: candidates per cast (< primitives: the filter dropped some),
  parked rays, parked rays the probe alone would have let through, rays that left a hull facet, leaf pre-tests.

The PT_DIAG build is a checker's build of the product kernels, not the product: it never runs outside these tests.
usage: diag_child.py SET   with SET in {configs, fullsize, fuzz, convex, rooms, wide}"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "raytracer.c_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def scene_sets(which):
    from rt_amd import scene as S
    from test_gpu_parity import _random_scene
    from util import whitted_scene, convex_body_scene
    if which == "configs":   # BASELINE configurations 1-5, reduced
        return [("config %d" % c, S.build_scene(c, w, h, spp)) for c, w, h, spp in
                [(1, 256, 256, 4), (2, 400, 300, 8), (3, 240, 136, 8), (4, 480, 270, 16), (5, 192, 108, 8)]]
    if which == "fullsize":  # configurations 2-4 at their own image sizes (the tile cones of tile_cull are those of the bench), few samples
        return [("config %d full size" % c, S.build_scene(c, samples=spp)) for c, spp in [(2, 4), (3, 2), (4, 2)]]
    if which == "fuzz":      # 6 random sphere scenes, 6 random mesh scenes (flat filter and hierarchy), the every-branch scene;
        # materials rotate so that the static (_refr), the pooled _chk and the plain kernel families all come up
        kinds = ["all", "no_glass", "plain"]
        sets = [("fuzz %d" % k, _random_scene(k, False, 0, materials=kinds[k % 3])) for k in range(6)]
        sets += [("fuzz mesh %d" % k, _random_scene(k, True, n, materials=kinds[k % 3]))
                 for k, n in zip(range(100, 106), [3, 40, 250, 300, 700, 2000])]
        from util import walls_scene
        sets += [("walls %d" % k, walls_scene(k, with_mesh=k >= 4)) for k in range(8)]
        return sets + [("whitted scene", whitted_scene())]
    if which == "convex":    # convex bodies at 64 spp: ~1e6 bounces off hull facets each, all walked
        sets = [("convex body %d" % k, convex_body_scene(k, 160, 100, 64)[0]) for k in range(4)]
        # ... and two of them turned to glass: BOTH children of a refractive hit leave the facet on the side the ray came from
        # (trace_step: the hull-facet rule for refraction children; pt_render_tiles_tri_queued_refr*)
        from rt_amd import abi
        for k in (4, 5):
            sc = convex_body_scene(k, 160, 100, 32)[0]
            sc.max_depth = 5
            for m in range(sc.n_meshes):
                sc.meshes[m].flags = abi.M_REFRACTION
            sets.append(("glass convex body %d" % k, sc))
        return sets
    if which == "rooms":     # rooms of more than 256 spheres (pt_render_tiles_pool_mem: geometry from memory, scalar-table filter)
        from util import packed_room
        from util import room_with_mesh   # ... and one with a mesh of 600 triangles: the parked-walk body with the spheres from memory
        return [("room %d" % n, packed_room(n, k, 160, 96, 4, 8)) for k, n in enumerate([249, 500, 1500])] + \
               [("room 300 + mesh", room_with_mesh(300, 3, 160, 96, 4, 8))]
    if which == "wide":      # the wider sweep of tools/diag_fuzz.py
        kinds = ["all", "no_glass", "plain"]
        sets = [("fuzz %d" % k, _random_scene(k, False, 0, materials=kinds[k % 3])) for k in range(40)]
        sets += [("fuzz mesh %d" % k, _random_scene(k, True, n, materials=kinds[k % 3])) for k, n in
                 zip(range(100, 112), [3, 10, 40, 120, 250, 300, 400, 700, 1000, 60, 500, 2000])]
        return sets + [("convex body %d" % k, convex_body_scene(k, 160, 100, 64)[0]) for k in range(8)]
    raise SystemExit(f"unknown scene set {which!r}")


def main():
    shim_path = os.environ.get("RT_HIP_SHIM_PATH", "")
    if "diag" not in os.path.basename(shim_path):
        raise SystemExit("run with RT_HIP_SHIM_PATH=<...>/librt_hip_diag.so (the PT_DIAG build)")
    os.environ["RT_HIP_DIAG_WALK_REJECTED"] = "1"   # read by the shim at every launch
    import torch
    from rt_amd import gpu as G
    for name, sc in scene_sets(sys.argv[1] if len(sys.argv) > 1 else "configs"):
        for integrator in ("path", "whitted"):
            if integrator == "whitted" and sc.max_depth > 32:
                continue
            gs = G.GpuScene(sc)
            stats = torch.zeros(48, dtype=torch.int64, device="cuda")
            try:
                gs.render_tiles(1666943821, 0, 1, G.n_tiles(sc.width, sc.height), stats=stats, integrator=integrator)
            except G.ShimError as e:
                print(json.dumps({"scene": name, "integrator": integrator, "skipped": str(e)}), flush=True)
                gs.close()
                continue
            torch.cuda.synchronize()
            st = stats.cpu().tolist()
            d = st[4:]
            print(json.dumps({"scene": name, "integrator": integrator, "kernel": gs.kernel_name(integrator),
                              "n_primitives": sc.n_primitives, "casts": st[1], "violations": d[12],
                              "candidates": d[3], "parked": d[17], "parked_probe_would_park": d[23],
                              "left_hull_facet": d[28], "tile_cannot_see_mesh": d[38], "leaf_pretests": d[16], "small_mesh_pretests": d[35],
                              "walked_found_triangle": d[18], "walls_pruned": d[37]}), flush=True)
            gs.close()


if __name__ == "__main__":
    main()
