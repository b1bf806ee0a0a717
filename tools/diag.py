#!/usr/bin/env python3
"""Wave- and lane-level event counts of the render kernels from the PT_DIAG build (make shim-diag).
usage: RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/librt_hip_diag.so python tools/diag.py [config] [spp] [--json profiles/diag_c<N>.json]
--json writes the lane-level events per ray-bounce that bench.py's executed-work model reads (EXEC_FLOPS), stamped with
the hash of the kernel sources they were counted on."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "raytracer.c_amd"))
json_out = None
if "--json" in sys.argv:
    k = sys.argv.index("--json")
    json_out = sys.argv[k + 1]
    del sys.argv[k:k + 2]
import torch
from rt_amd import gpu as G, scene as S
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sc = S.build_scene(cfg, samples=spp)
gs = G.GpuScene(sc)
stats = torch.zeros(48, dtype=torch.int64, device="cuda")
total = G.n_tiles(sc.width, sc.height)
gs.render_tiles(1666943821, 0, 1, total, stats=stats)
torch.cuda.synchronize()
st = stats.cpu().tolist()
rays, casts = st[0], st[1]
d = st[4:]
names = ["loop iters (wave)", "loop lanes", "phase2 iters (wave)", "phase2 cands (lane)", "sqrt blocks (wave)", "sqrt lanes",
         "fresh blocks (wave)", "fresh lanes", "hit blocks (wave)", "hit lanes", "reject-loop iters (wave)", "reject-loop lanes", "FILTER VIOLATIONS (must be 0)", "bvh node visits (wave)", "bvh leaf triangle tests (wave)", "bvh node-visit lanes", "bvh leaf pre-tests (wave)", "parked rays", "walked rays returning a triangle",
         "walks of 1 node visit", "walks of 2-3", "walks of 4-6", "walks of 7+",
         "parked rays inside the triangles' bounding sphere",
         "parked from inside the ball, found a triangle", "parked from inside the ball, found none",
         "parked from outside the ball, found a triangle", "parked from outside the ball, found none",
         "rays leaving a hull facet (no probe)", "phase-2 iters if each lane kept one wall (wave)", "phase-2 iters of non-wall candidates alone (wave)",
         "phase-2 iters of wall candidates alone (wave)", "wall candidates (lane)", "non-wall candidates (lane)", "small-mesh fp32 pre-test iters (wave)",
         "small-mesh fp32 pre-tests (lane)", "small-mesh exact triangle iters (wave)", "wall-sized spheres pruned before the exact tests (lane)",
         "camera rays of tiles that cannot see the mesh (no probe)", "filter evaluations (lane x primitive)", "bvh leaf pre-tests (lane)",
         "exact triangle tests (lane)", "mesh probe evaluations (lane)", "exact sphere tests (lane)"]
for n, v in zip(names, d):
    print(f"{n:28s} {v:15d}")
assert d[12] == 0, 'the conservative filter dropped a sphere the exact test accepts'
it = d[0]
print(f"rays {rays}  casts {casts}  rays/(64*iters) = lane occupancy of the loop: {rays / (64.0 * it):.3f}")
print(f"phase-2 iterations per loop iteration: {d[2] / it:.2f}; candidates per ray: {d[3] / casts:.2f}; "
      f"phase-2 lane occupancy {d[3] / (64.0 * d[2]):.3f}")

print(f"fresh per iter {d[6] / it:.3f} lanes/64 {d[7] / (64.0 * max(d[6], 1)):.3f}")
print(f"hit per iter {d[8] / it:.3f} lanes/64 {d[9] / (64.0 * max(d[8], 1)):.3f}")
print(f"bvh node-visit wave iterations per loop iter {d[13] / it:.1f} (lanes/64 {d[15] / (64.0 * max(d[13], 1)):.3f}, "
      f"node visits per cast {d[15] / max(casts, 1):.1f}); leaf triangle-test wave iterations per loop iter {d[14] / it:.1f}, "
      f"fp32 pre-tests before them {d[16] / it:.1f}")
if d[17]:
    if os.environ.get("RT_HIP_DIAG_WALK_REJECTED") == "1":
        print(f"(RT_HIP_DIAG_WALK_REJECTED=1: every ray that passes the boxes is parked; the shipped build parks only those also "
              f"inside the bounding sphere: {d[23] / d[17]:.3f} of them)")
    print(f"parked rays per cast {d[17] / casts:.3f}; of the walked rays {d[18] / d[17]:.3f} return with a triangle; walks by node "
          f"visits 1: {d[19] / d[17]:.3f}  2-3: {d[20] / d[17]:.3f}  4-6: {d[21] / d[17]:.3f}  7+: {d[22] / d[17]:.3f}")
if d[29]:
    print(f"sphere candidates per ray: walls (r > 1000) {d[32] / casts:.2f}, others {d[33] / casts:.2f}; phase-2 iterations per trip: "
          f"now {d[2] / it:.2f}, walls alone {d[31] / it:.2f}, others alone {d[30] / it:.2f}, with one wall per lane {d[29] / it:.2f}")
if d[34]:
    print(f"small mesh: fp32 pre-test wave iterations per trip {d[34] / it:.2f} for {d[35] / casts:.2f} bounding-sphere candidates per ray; "
          f"exact triangle wave iterations per trip {d[36] / it:.2f}, exact sphere {(d[2] - d[36]) / it:.2f}")
print(f"reject iters per loop iter {d[10] / it:.2f} lanes/64 {d[11] / (64.0 * max(d[10], 1)):.3f}")

print(f"per ray-bounce: filter evaluations {d[39] / casts:.2f}, exact sphere tests {d[43] / casts:.3f}, exact triangle tests {d[41] / casts:.4f}, "
      f"node visits {d[15] / casts:.3f}, leaf pre-tests {d[40] / casts:.3f}, probes {d[42] / casts:.3f}, rejection rounds {d[11] / casts:.3f}")
if json_out:
    sys.path.insert(0, ROOT)
    from bench import kernel_source_sha256
    # a direction is sampled once per diffuse hit that survives the roulette: rounds / 1.91 would be a model; count it as
    # the lanes that leave the rejection loop = hits that go on diffusely ~ (casts - ended paths) minus mirrors: not counted
    # apart, so the rounds' own lane count is used for both (an accepted sample is the last round of its lane)
    per = {"camera_sample": d[7] / casts, "exact_sphere": d[43] / casts, "exact_triangle": (d[41]) / casts,
           "hit": d[9] / casts, "rejection_round": d[11] / casts, "direction": d[11] / casts / 1.9099,
           "filter_sphere": d[39] / casts, "node_visit": d[15] / casts, "leaf_pretest": (d[40] + d[35]) / casts,
           "probe": d[42] / casts}
    json.dump({"config": cfg, "width": sc.width, "height": sc.height, "spp": spp, "kernel": gs.kernel_name(),
               "source_sha256": kernel_source_sha256(), "rays": rays, "ray_bounces": casts,
               "per_ray_bounce": per,
               "note": "lane-level events of one PT_DIAG frame divided by its ray-bounces; 'direction' = rejection rounds / 1.9099 "
                       "(the mean rounds per accepted sample, 1 / (pi / 6) -- accepted samples are not counted apart); "
                       "'leaf_pretest' includes the small-mesh kernels' per-lane fp32 pre-tests",
               "raw": {n: v for n, v in zip(names, d)}}, open(json_out, "w"), indent=1)
    print("wrote", json_out)
