"""Pins the oracle: oracle/pt_oracle.c (our CPU restatement) must be BIT-IDENTICAL to the
reference's own compiled trace_path()/intersect()/primitives (oracle/_ref, built in place
from /root/reference).  Skipped where the compiled reference is absent (it travels to the
GPU box as a prebuilt .so; the golden fixtures in test_golden.py cover the same ground
everywhere)."""
import ctypes as C

import numpy as np
import pytest

from conftest import SEED


def _scene(cfg, w, h, spp):
    from rt_amd import scene as S
    return S.build_scene(cfg, w, h, spp)


@pytest.mark.parametrize("cfg,w,h,spp", [(1, 256, 256, 4), (2, 200, 150, 8), (4, 160, 90, 8), (4, 64, 36, 64)])
def test_frames_bit_identical(pt, ref, cfg, w, h, spp):
    sc = _scene(cfg, w, h, spp)
    m1, b1, s1 = pt.render_pixels(sc, SEED)
    m2, b2, s2 = ref(sc.max_depth).render_pixels(sc, SEED)
    assert np.array_equal(m1, m2), "linear fp64 means differ from the compiled reference"
    assert np.array_equal(b1, b2)
    assert s1["rays"] == s2["rays"] and s1["tests"] == s2["tests"]
    assert s1["tests"] == s1["casts"] * sc.n_objects


def test_depth_variants(pt, ref):
    """one scene at every MAX_DEPTH the compiled reference was built for"""
    rays = []
    for d in (4, 5, 8, 16):
        from rt_amd import scene as S
        sc = S.build_scene(4, 96, 54, 4, max_depth=d)
        m1, _, s1 = pt.render_pixels(sc, SEED)
        m2, _, s2 = ref(d).render_pixels(sc, SEED)
        assert np.array_equal(m1, m2) and s1["rays"] == s2["rays"]
        rays.append(s1["rays"])
    assert rays == sorted(rays) and rays[0] < rays[-1]


def test_per_sample_traces(pt, ref):
    sc = _scene(4, 320, 180, 32)
    r = ref(sc.max_depth)
    rng = np.random.default_rng(3)
    for _ in range(200):
        x, y, s = int(rng.integers(0, 320)), int(rng.integers(0, 180)), int(rng.integers(0, 32))
        c1, s1 = pt.trace_sample(sc, x, y, s, SEED)
        c2, s2 = r.trace_sample(sc, x, y, s, SEED)
        assert np.array_equal(c1, c2)
        assert (s1["rays"], s1["tests"], s1["draws"]) == (s2["rays"], s2["tests"], s2["draws"])


@pytest.mark.parametrize("cfg,w,h,spp", [(1, 128, 128, 2), (2, 160, 120, 2), (4, 160, 90, 2)])
def test_whitted_frames_bit_identical(pt, ref, cfg, w, h, spp):
    """cast_ray (raytracer.c:556-641), compiled in the reference's TU though render() does not
    call it as shipped: the restatement must match it bit for bit as well"""
    sc = _scene(cfg, w, h, spp)
    m1, b1, s1 = pt.render_pixels(sc, SEED, integrator="whitted")
    m2, b2, s2 = ref(sc.max_depth).render_pixels(sc, SEED, integrator="whitted")
    assert np.array_equal(m1, m2) and np.array_equal(b1, b2)
    assert s1["rays"] == s2["rays"] and s1["tests"] == s2["tests"]
    assert s1["tests"] == s1["casts"] * sc.n_objects and s1["draws"] == 2 * w * h * spp


def test_whitted_every_branch_bit_identical(pt, ref):
    from util import whitted_scene
    sc = whitted_scene(samples=3)
    m1, b1, s1 = pt.render_pixels(sc, SEED, integrator="whitted")
    m2, b2, s2 = ref(sc.max_depth).render_pixels(sc, SEED, integrator="whitted")
    assert np.array_equal(m1, m2) and np.array_equal(b1, b2)
    assert (s1["rays"], s1["tests"]) == (s2["rays"], s2["tests"])
    # the path tracer is unaffected by the switch having been used on the same library
    m3, _, s3 = ref(sc.max_depth).render_pixels(sc, SEED)
    m4, _, s4 = pt.render_pixels(sc, SEED)
    assert np.array_equal(m3, m4) and s3["rays"] == s4["rays"] and not np.array_equal(m3, m2)


def test_restated_pixel_loop_equals_render_as_shipped(ref):
    """SURVEY 8c check 1: the only restated lines of the harness (the pixel loop of render(),
    raytracer.c:197-221) reproduce the reference's render() byte for byte when both draw
    from libc rand() (single thread)."""
    sc = _scene(1, 96, 64, 4)
    r = ref(4)
    a, sa = r.render_as_shipped(sc, 1666943821)
    b, sb = r.render_loop_libc(sc, 1666943821)
    assert np.array_equal(a, b) and sa == sb and a.any()


def test_primitives_bit_identical(pt, ref):
    r = ref(5)
    rng = np.random.default_rng(11)
    for _ in range(500):
        o = rng.uniform(-30, 30, 3)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        ray = np.concatenate([o, d])
        c, rad = rng.uniform(-30, 30, 3), rng.uniform(0.5, 15)
        assert pt.intersect_sphere(ray, c, rad) == r.intersect_sphere(ray, c, rad)
        verts = rng.uniform(-10, 10, 15)
        h1, t1 = pt.intersect_triangle(ray, verts)
        h2, t2 = r.intersect_triangle(ray, verts)
        assert h1 == h2 and (not h1 or np.array_equal(t1, t2))
        v9 = rng.uniform(-5, 5, 9)
        assert np.array_equal(pt.surface_normal(v9), r.surface_normal(v9))
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        assert np.array_equal(pt.reflect(d, n), r.reflect(d, n))
        assert np.array_equal(pt.refract(d, n, 1.0), r.refract(d, n, 1.0))
        u, v = rng.uniform(0, 1, 2)
        assert np.array_equal(pt.checkered(n * n, u, v, 1e5), r.checkered(n * n, u, v, 1e5))


def test_scene_scan_and_camera(pt, ref):
    sc = _scene(4, 160, 90, 1)
    r = ref(5)
    rng = np.random.default_rng(12)
    cam1 = pt.init_camera((0, 0, 50), (0, 0, 0), 160, 90)
    cam2 = r.init_camera((0, 0, 50), (0, 0, 0), 160, 90)
    assert bytes(cam1) == bytes(cam2) == bytes(sc.camera)
    for _ in range(300):
        u, v = rng.uniform(0, 1, 2)
        ray = pt.camera_ray(cam1, u, v)
        assert np.array_equal(ray, r.camera_ray(cam2, u, v))
        ok1, pn1, tuv1, id1 = pt.intersect_scene(ray, sc.objects, sc.n_objects)
        ok2, pn2, tuv2, id2 = r.intersect_scene(ray, sc.objects, sc.n_objects)
        assert ok1 == ok2 and ok1  # closed room: always a hit
        assert id1 == id2 and np.array_equal(pn1, pn2)
        assert np.array_equal(tuv1[1:], tuv2[1:])  # u, v (hit.t of the reference is stale by design)


def test_struct_layout_matches_compiled_reference(ref):
    from rt_amd import abi
    lay = ref(5).layout()
    assert lay[0] == C.sizeof(abi.Object) == 88
    assert lay[1:6] == [abi.Object.flags.offset, abi.Object.radius.offset, abi.Object.center.offset,
                        abi.Object.color.offset, abi.Object.emission.offset]
    assert lay[6] == C.sizeof(abi.Camera) == 96
    assert lay[7] == C.sizeof(abi.Options) == 56
    assert lay[8] == C.sizeof(abi.Ray) == 48
    assert lay[9] == C.sizeof(abi.Hit) == 80
    assert lay[10] == C.sizeof(abi.Vertex) == 40
    assert lay[11] == C.sizeof(abi.Vec3) == 24
    assert lay[12:15] == [abi.Options.width.offset, abi.Options.height.offset, abi.Options.samples.offset]
    assert lay[15] == C.sizeof(abi.TriangleMesh) == 16


# ---- the mesh scan (raytracer.c:417-435, a comment block in the reference) pinned IN COMPOSITION:
# ---- revived around the reference's compiled primitives and called by its compiled trace_path()

@pytest.mark.parametrize("cfg,w,h,spp", [(2, 120, 90, 4), (4, 96, 54, 4)])
def test_mesh_hook_is_transparent_on_sphere_scenes(ref, ref_mesh, cfg, w, h, spp):
    """the hooked build must reproduce the unhooked compiled reference bit for bit (both integrators)"""
    sc = _scene(cfg, w, h, spp)
    for integ in ("path", "whitted"):
        m1, b1, s1 = ref(sc.max_depth).render_pixels(sc, SEED, integrator=integ)
        m2, b2, s2 = ref_mesh(sc.max_depth).render_pixels(sc, SEED, integrator=integ)
        assert np.array_equal(m1, m2) and np.array_equal(b1, b2) and s1 == s2


def test_revived_sphere_branch_equals_live_intersect(ref_mesh):
    """the sphere branch of the revived loop is the live text :404-411; its result must equal the
    reference's compiled intersect() (harness_isect_2) field by field, stale hit.t included"""
    sc = _scene(4, 160, 90, 1)
    r = ref_mesh(16)
    rng = np.random.default_rng(21)
    for _ in range(2000):
        o = rng.uniform(-15, 15, 3)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        ray = np.concatenate([o, d])
        ok, pn, tuv, oid = r.intersect_scene(ray, sc.objects, sc.n_objects)
        m = r.intersect_mesh_scene(ray, sc)
        assert ok == m["hit"]
        if ok:
            assert oid == m["id"] and np.array_equal(pn, np.concatenate([m["point"], m["normal"]]))
            assert np.array_equal(tuv, [m["t_stale"], m["u"], m["v"]])


@pytest.mark.parametrize("cfg,w,h,spp", [(3, 160, 90, 8), (5, 48, 27, 2)])
def test_mesh_frames_bit_identical(pt, ref_mesh, cfg, w, h, spp):
    """configs 3 and 5 (BASELINE.json): whole frames out of the reference's compiled trace_path() +
    revived mesh scan == the restatement, bit for bit, counters included"""
    sc = _scene(cfg, w, h, spp)
    m1, b1, s1 = pt.render_pixels(sc, SEED)
    m2, b2, s2 = ref_mesh(sc.max_depth).render_pixels(sc, SEED)
    assert np.array_equal(m1, m2), "linear fp64 means differ from the compiled reference + revived mesh scan"
    assert np.array_equal(b1, b2)
    assert s1["rays"] == s2["rays"] and s1["tests"] == s2["tests"]
    assert s1["tests"] == s1["casts"] * (sc.n_objects + sc.n_triangles)


@pytest.mark.parametrize("integ", ["path", "whitted"])
def test_mesh_soup_frames_bit_identical(pt, ref_mesh, integ):
    """random triangle soup with texture coordinates, duplicated and degenerate triangles, checkered
    materials (stale hit.u / hit.v of the literal scan), two meshes, a mirror: both integrators"""
    from util import mesh_soup_scene
    sc = mesh_soup_scene()
    m1, b1, s1 = pt.render_pixels(sc, SEED, integrator=integ)
    m2, b2, s2 = ref_mesh(sc.max_depth).render_pixels(sc, SEED, integrator=integ)
    assert np.array_equal(m1, m2) and np.array_equal(b1, b2)
    assert s1["rays"] == s2["rays"] and s1["tests"] == s2["tests"]


def test_mesh_per_sample_traces(pt, ref_mesh):
    sc = _scene(3, 320, 180, 16)
    r = ref_mesh(sc.max_depth)
    rng = np.random.default_rng(4)
    for _ in range(200):
        x, y, s = int(rng.integers(0, 320)), int(rng.integers(0, 180)), int(rng.integers(0, 16))
        c1, s1 = pt.trace_sample(sc, x, y, s, SEED)
        c2, s2 = r.trace_sample(sc, x, y, s, SEED)
        assert np.array_equal(c1, c2)
        assert (s1["rays"], s1["tests"], s1["draws"]) == (s2["rays"], s2["tests"], s2["draws"])


def test_mesh_closest_hit_on_random_rays(pt, ref_mesh):
    """>= 10,000 random rays against spheres + two triangle soups (duplicates, degenerates): hit flag,
    object id, point, normal, u, v and the number of primitive tests bit-identical to the revived
    scan around the compiled primitives.  Also shows the literal scan's quirk is exercised: on some
    rays hit.u / hit.v are NOT the closest primitive's own (ref_harness.c)."""
    from util import mesh_soup_scene
    sc = mesh_soup_scene(seed=9, n_tris=60)
    r = ref_mesh(8)
    rng = np.random.default_rng(31)
    n_hit = n_tri_win = n_stale = 0
    for k in range(10000):
        if k % 3 == 0:   # towards the soup from outside
            o = rng.normal(size=3)
            o *= 25 / np.linalg.norm(o)
            d = rng.uniform(-6, 6, 3) - o
        else:            # from inside it
            o = rng.uniform(-8, 8, 3)
            d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        ray = np.concatenate([o, d])
        a = pt.intersect_mesh_scene(ray, sc)
        b = r.intersect_mesh_scene(ray, sc)
        assert a["hit"] == b["hit"] and a["tests"] == b["tests"] == sc.n_objects + sc.n_triangles
        if not a["hit"]:
            continue
        n_hit += 1
        assert a["id"] == b["id"] and a["min_t"] == b["min_t"]
        assert np.array_equal(a["point"], b["point"]) and np.array_equal(a["normal"], b["normal"])
        assert (a["u"], a["v"]) == (b["u"], b["v"])
        n_tri_win += a["id"] >= sc.n_objects
        n_stale += (b["u"], b["v"]) != (b["u_win"], b["v_win"])
    assert n_hit > 5000 and n_tri_win > 500 and n_stale > 100, (n_hit, n_tri_win, n_stale)


def test_mesh_layout_matches_compiled_reference(ref_mesh):
    from rt_amd import abi
    r = ref_mesh(5)
    assert r.lib.ref_mesh_layout() == C.sizeof(abi.MeshObject) == 72
