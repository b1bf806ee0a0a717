"""The exhaustive checks of the kernels' CONSERVATIVE rules, in the driver's suite.

The render kernels may skip exact fp64 tests only through rules that can never change an outcome: the packed-fp32
phase-1 filter (three forms), the per-tile culling of the primary trips' filter (tile_cull: primitives no camera ray of the
tile can reach are not even looked at), the pruning of wall-sized spheres among themselves by fp32 distance bounds (BigPrune), the per-lane fp32 Moeller-Trumbore
pre-test (small meshes and hierarchy leaves), the bounding-sphere probe in front of a hierarchy walk, and the hull-facet rule (a bounce that leaves a convex facet
on its outer side is not walked).  Each rests on a hand-derived error bound (pt_kernel.hip: pt_build_filter; pt_filter.h, pt_intersect.h:
tri_may_hit32; rt_hip_shim.hip: mesh_bound_for, hull_margin_for).  The PT_DIAG build of the same kernels
(`make shim-diag`, part of `make all`: raytracer.c_amd/csrc/librt_hip_diag.so) re-checks every application of every
rule at run time: each primitive the filter or a pre-test drops is put through the exact test, and with
RT_HIP_DIAG_WALK_REJECTED=1 every ray the probe or the hull rule would not walk is walked all the same; anything
found counts into stats[4 + 12].  That count must be ZERO.

The diag library is loaded in child processes (RT_HIP_SHIM_PATH), never into the test process, whose shim is the
shipped one."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG_LIB = os.path.join(ROOT, "raytracer.c_amd", "csrc", "librt_hip_diag.so")


def _run(which, timeout=600):
    assert os.path.exists(DIAG_LIB), "make all builds librt_hip_diag.so"
    env = dict(os.environ, RT_HIP_SHIM_PATH=DIAG_LIB, RT_HIP_DIAG_WALK_REJECTED="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag_child.py"), which], env=env, capture_output=True,
                       text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    recs = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert recs, p.stdout[-2000:]
    return recs


def _no_violations(recs):
    bad = [r for r in recs if r.get("violations", 0) != 0]
    assert not bad, f"a conservative rule dropped something the exact test accepts: {bad}"


def test_diag_configs_1_to_5_zero_violations():
    recs = _run("configs")
    _no_violations(recs)
    path = {r["scene"]: r for r in recs if r["integrator"] == "path" and "skipped" not in r}
    assert set(path) == {"config %d" % c for c in range(1, 6)}
    for r in path.values():   # the filter really dropped primitives (else the re-check had nothing to check)
        assert 0 < r["candidates"] < r["casts"] * r["n_primitives"], r
    # config 4's six walls are pruned among themselves before the exact tests: that happened, and every pruned wall lost strictly
    assert path["config 4"]["walls_pruned"] > path["config 4"]["casts"] // 2
    c3, c5 = path["config 3"], path["config 5"]
    assert c3["kernel"] == "pt_render_tiles_tri" and c3["small_mesh_pretests"] > 0
    assert c5["kernel"] == "pt_render_tiles_tri_queued_sph"
    # rays the probe's bounding sphere rejects were walked too (parked > what the shipped build parks), bounces off hull
    # facets were seen, leaves were pre-tested, and walks did find triangles
    assert c5["parked"] > c5["parked_probe_would_park"] > 0 and c5["left_hull_facet"] > 0
    # camera rays of tiles whose cone cannot reach the mesh's bounding ball go without a probe: such tiles were seen (and,
    # walked all the same here, their rays found nothing: no violations above)
    assert c5["tile_cannot_see_mesh"] > 0
    assert c5["leaf_pretests"] > 0 and c5["walked_found_triangle"] > 0


def test_diag_fuzz_scenes_zero_violations():
    recs = _run("fuzz")
    _no_violations(recs)
    done = [r for r in recs if "skipped" not in r]
    assert len({r["scene"] for r in done}) == 21 and {r["integrator"] for r in done} == {"path", "whitted"}
    walls = [r for r in done if r["scene"].startswith("walls") and r["integrator"] == "path"]
    assert len(walls) == 8 and sum(r["walls_pruned"] > 0 for r in walls) >= 6, walls
    kernels = {r["kernel"] for r in done}
    # the static (_refr), pooled (plain / _chk), small-mesh, parked-walk and cast_ray families were all exercised
    assert any(k.startswith("pt_render_tiles_tri_queued") for k in kernels), kernels
    assert any(k.startswith("pt_whitted_tiles") for k in kernels), kernels
    assert any("_refr" in k for k in kernels) and any(k in ("pt_render_tiles", "pt_render_tiles_chk") for k in kernels), kernels
    assert "pt_render_tiles_refr_pool" in kernels, kernels   # small glass scenes: the pooled body (round 4)
    assert any(k in ("pt_render_tiles_tri", "pt_render_tiles_tri_chk") for k in kernels), kernels


def test_diag_convex_bodies_zero_violations():
    recs = _run("convex")
    _no_violations(recs)
    path = [r for r in recs if r["integrator"] == "path" and "skipped" not in r]
    assert len(path) == 6
    for r in path:   # thousands of bounces off hull facets each -- all walked under RT_HIP_DIAG_WALK_REJECTED, none found a triangle
        assert r["left_hull_facet"] > 1000 and r["parked"] > 0, r
    # the two glass bodies: the parked-walk body's refraction form, whose refraction children (both of a hit) take the rule too
    glass = [r for r in path if r["scene"].startswith("glass")]
    assert len(glass) == 2 and all(r["kernel"].startswith("pt_render_tiles_tri_queued_refr") for r in glass), glass


def test_diag_full_size_tile_cones_zero_violations():
    """configurations 2, 3, 4 at their own image sizes: the per-tile cones of the primary trips are the bench's"""
    recs = _run("fullsize")
    _no_violations(recs)
    path = [r for r in recs if r["integrator"] == "path" and "skipped" not in r]
    assert {r["kernel"] for r in path} == {"pt_render_tiles", "pt_render_tiles_tri"} and len(path) == 3


def test_diag_rooms_beyond_the_staging_budget_zero_violations():
    """rooms of 257 .. 1,508 spheres: the pooled body with geometry from memory and the filter table by scalar loads"""
    recs = _run("rooms")
    _no_violations(recs)
    path = [r for r in recs if r["integrator"] == "path" and "skipped" not in r]
    mesh = [r for r in path if r["scene"].endswith("mesh")]
    path = [r for r in path if not r["scene"].endswith("mesh")]
    assert len(path) == 3 and {r["kernel"] for r in path} == {"pt_render_tiles_pool_mem_s"}
    assert all(r["walls_pruned"] > 0 for r in path)   # the six walls lead these rooms too: pruned among themselves here as well
    for r in path:
        assert 0 < r["candidates"] < r["casts"] * r["n_primitives"] // 4, r
    # the room with a mesh: parked walks with the spheres' sign-form filter pairs read from memory -- the filter dropped spheres,
    # rays were parked and walked (also those the probe rejects), leaves were pre-tested: 0 violations above
    assert len(mesh) == 1 and mesh[0]["kernel"] == "pt_render_tiles_tri_queued_mem", mesh
    assert 0 < mesh[0]["candidates"] < mesh[0]["casts"] * 308 and mesh[0]["parked"] > mesh[0]["parked_probe_would_park"] > 0 and mesh[0]["leaf_pretests"] > 0
