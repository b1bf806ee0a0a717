"""The kernel pick table (pt_kernel.hip: pt_pick_table) pinned row by row, WITHOUT a GPU: rt_hip_kernel_for_class() is the
table itself behind the C-ABI.  Every member of the shipped family restates the same two functions of the reference --
trace_path (raytracer.c:482-554) or cast_ray (:556-641) -- so which member a scene takes must never change a result, only
its speed; what must not happen silently is a scene class changing rows.  The expectations below are
profiles/r04_kernel_pick_table.txt's default column (the scene classes of tools/kernel_pick_table.py) plus every fallback
row, and the closing test walks the whole class space to show that the table is total and that no row is dead.
"""
import ctypes as C
import itertools

import pytest

PATH, CAST = 0, 1


class SceneClass(C.Structure):
    _fields_ = [("integrator", C.c_uint32), ("n_spheres", C.c_uint32), ("n_meshes", C.c_uint32), ("n_triangles", C.c_uint32),
                ("any_checker", C.c_uint32), ("any_refract", C.c_uint32), ("any_mirror_glass", C.c_uint32),
                ("wide_range", C.c_uint32), ("mesh_round", C.c_uint32), ("samples_per_chunk", C.c_int32),
                ("max_depth", C.c_int32), ("have_park_ws", C.c_uint32), ("wide_pend_ok", C.c_uint32)]


@pytest.fixture(scope="module")
def pick():
    from rt_amd import abi
    shim = abi.load_shim()

    def f(integrator=PATH, spheres=0, meshes=0, tris=0, chk=0, refr=0, glass2=0, wide=0, round_=0, spp=64, depth=5,
          park_ws=1, wide_pend=1):
        c = SceneClass(integrator, spheres, meshes, tris, chk, refr, glass2, wide, round_, spp, depth, park_ws, wide_pend)
        return shim.rt_hip_kernel_for_class(C.byref(c)).decode()
    return f


# (what, class, trace_path kernel, cast_ray kernel) -- profiles/r04_kernel_pick_table.txt, default column
R04_ROWS = [
    ("config 1", dict(spheres=4), "pt_render_tiles", "pt_whitted_tiles"),
    ("config 2", dict(spheres=10), "pt_render_tiles", "pt_whitted_tiles"),
    ("config 3", dict(spheres=5, meshes=1, tris=12), "pt_render_tiles_tri", "pt_whitted_tiles_tri"),
    ("config 4", dict(spheres=38), "pt_render_tiles", "pt_whitted_tiles"),
    ("config 5", dict(spheres=8, meshes=1, tris=10240, round_=1), "pt_render_tiles_tri_queued_sph", "pt_whitted_tiles_tri_big"),
    ("glass spheres", dict(spheres=5, refr=1), "pt_render_tiles_refr_pool", "pt_whitted_tiles"),
    ("config 5, glass mesh", dict(spheres=8, meshes=1, tris=10240, round_=1, refr=1), "pt_render_tiles_tri_queued_refr_sph",
     "pt_whitted_tiles_tri_big"),
    ("config 5, checkered wall", dict(spheres=8, meshes=1, tris=10240, round_=1, chk=1), "pt_render_tiles_tri_queued_chk_sph",
     "pt_whitted_tiles_tri_big"),
    ("config 3, glass cube", dict(spheres=5, meshes=1, tris=12, refr=1), "pt_render_tiles_tri_refr_pool", "pt_whitted_tiles_tri"),
    ("room of 120 spheres", dict(spheres=126), "pt_render_tiles_pool_mem_s", "pt_whitted_tiles"),
    ("room of 300 spheres", dict(spheres=306), "pt_render_tiles_pool_mem_s", "pt_whitted_tiles_mem"),
    ("room of 300, glass", dict(spheres=306, refr=1), "pt_render_tiles_refr_pool_mem", "pt_whitted_tiles_mem"),
    ("room of 300 + 600 triangles", dict(spheres=306, meshes=1, tris=600), "pt_render_tiles_tri_queued_mem", "pt_whitted_tiles_mem"),
]


@pytest.mark.parametrize("what,cls,path_kernel,cast_kernel", R04_ROWS, ids=[r[0] for r in R04_ROWS])
def test_the_scene_classes_of_round_4_keep_their_rows(pick, what, cls, path_kernel, cast_kernel):
    assert pick(PATH, **cls) == path_kernel
    assert pick(CAST, **cls) == cast_kernel


# the rest of the table: every other member, and every fallback, by the class that reaches it
OTHER_ROWS = [
    # hierarchy scenes by probe form and material
    (dict(spheres=8, meshes=1, tris=10240), "pt_render_tiles_tri_queued"),
    (dict(spheres=8, meshes=1, tris=10240, chk=1), "pt_render_tiles_tri_queued_chk"),
    (dict(spheres=8, meshes=1, tris=10240, refr=1), "pt_render_tiles_tri_queued_refr"),
    # ... without the ring workspace (its allocation failed): the lane-waiting kernels
    (dict(spheres=8, meshes=1, tris=10240, round_=1, park_ws=0), "pt_render_tiles_tri_big"),
    (dict(spheres=8, meshes=1, tris=10240, chk=1, park_ws=0), "pt_render_tiles_tri_big_chk"),
    (dict(spheres=8, meshes=1, tris=10240, refr=1, park_ws=0), "pt_render_tiles_tri_big_refr"),
    # ... beyond fp32's comfortable range: no parked walks either
    (dict(spheres=8, meshes=1, tris=10240, wide=1), "pt_render_tiles_tri_big"),
    # ... with M_REFRACTION when the pending-ray pool could only be had narrow, or the windowed sums do not fit
    (dict(spheres=8, meshes=1, tris=10240, refr=1, wide_pend=0), "pt_render_tiles_tri_big_refr"),
    (dict(spheres=8, meshes=1, tris=10240, refr=1, spp=64, depth=30), "pt_render_tiles_tri_big_refr"),
    (dict(spheres=8, meshes=1, tris=10240, refr=1, spp=8193, depth=16), "pt_render_tiles_tri_big_refr"),   # per sample chunk: 8193 x 2^17 > 2^30
    (dict(spheres=8, meshes=1, tris=10240, refr=1, spp=8192, depth=16), "pt_render_tiles_tri_queued_refr"),
    (dict(spheres=8, meshes=1, tris=10240, refr=1, spp=1, depth=29), "pt_render_tiles_tri_queued_refr"),      # one sample per chunk at the deepest depth that fits
    # a mesh of few triangles next to more than 256 primitives: hierarchy, parked
    (dict(spheres=200, meshes=1, tris=100), "pt_render_tiles_tri_queued"),
    # small scenes by material
    (dict(spheres=10, chk=1), "pt_render_tiles_chk"),
    (dict(spheres=5, meshes=1, tris=12, chk=1), "pt_render_tiles_tri_chk"),
    (dict(spheres=5, refr=1, chk=1), "pt_render_tiles_refr_pool"),
    # ... whose windowed sums do not fit: the static kernels
    (dict(spheres=5, refr=1, depth=30), "pt_render_tiles_refr"),
    (dict(spheres=5, meshes=1, tris=12, refr=1, depth=30), "pt_render_tiles_tri_refr"),
    (dict(spheres=126, refr=1, depth=30), "pt_render_tiles_refr"),
    (dict(spheres=306, refr=1, depth=30), "pt_render_tiles_mem"),
    # small scenes out of range: the NaN-safe compare filter by scalar loads
    (dict(spheres=10, wide=1), "pt_render_tiles_big"),
    (dict(spheres=10, wide=1, chk=1), "pt_render_tiles_big_chk"),
    (dict(spheres=10, wide=1, refr=1), "pt_render_tiles_big_refr"),
    (dict(spheres=126, wide=1), "pt_render_tiles_big"),
    # sphere scenes that stream, by material
    (dict(spheres=126, chk=1), "pt_render_tiles_pool_mem_s_chk"),
    (dict(spheres=126, refr=1), "pt_render_tiles_refr_pool_mem"),
    (dict(spheres=306, chk=1), "pt_render_tiles_pool_mem_s_chk"),
    # scenes beyond the staging budget: out of range, with a small mesh, with a big one but no workspace, with glass and a mesh
    (dict(spheres=306, wide=1), "pt_render_tiles_pool_mem"),
    (dict(spheres=306, wide=1, chk=1), "pt_render_tiles_pool_mem_chk"),
    (dict(spheres=306, meshes=1, tris=100), "pt_render_tiles_pool_mem_tri"),
    (dict(spheres=306, meshes=1, tris=100, chk=1), "pt_render_tiles_pool_mem_tri_chk"),
    (dict(spheres=306, meshes=1, tris=600, chk=1), "pt_render_tiles_tri_queued_mem_chk"),
    (dict(spheres=306, meshes=1, tris=600, park_ws=0), "pt_render_tiles_pool_mem_tri"),
    (dict(spheres=306, meshes=1, tris=600, wide=1), "pt_render_tiles_pool_mem_tri"),
    (dict(spheres=306, meshes=1, tris=600, refr=1), "pt_render_tiles_mem"),
    (dict(spheres=306, refr=1, wide=1), "pt_render_tiles_mem"),
]


@pytest.mark.parametrize("cls,kernel", OTHER_ROWS, ids=[f"{k}:{'+'.join(f'{a}={b}' for a, b in c.items())}" for c, k in OTHER_ROWS])
def test_every_other_row_and_fallback(pick, cls, kernel):
    assert pick(PATH, **cls) == kernel


CAST_ROWS = [
    (dict(spheres=10, wide=1), "pt_whitted_tiles_big"),
    (dict(spheres=10, glass2=1), "pt_whitted_tiles_mem"),
    (dict(spheres=5, meshes=1, tris=12, glass2=1), "pt_whitted_tiles_mem"),
    (dict(spheres=126), "pt_whitted_tiles"),                      # cast_ray never streams by preference
    (dict(spheres=200, meshes=1, tris=100), "pt_whitted_tiles_tri_big"),
    (dict(spheres=5, refr=1, depth=40), "pt_whitted_tiles"),      # M_REFRACTION alone is one child under cast_ray: no limit
]


@pytest.mark.parametrize("cls,kernel", CAST_ROWS, ids=[k + ":" + "+".join(f"{a}={b}" for a, b in c.items()) for c, k in CAST_ROWS])
def test_cast_ray_rows(pick, cls, kernel):
    assert pick(CAST, **cls) == kernel


def test_the_table_is_total_and_every_shipped_kernel_is_some_class_s_row(pick):
    """walk the class space: every class gets a kernel of the family (the table has no gap), and every member of the shipped
    family is the row of at least one class (no dead row, nothing reachable only through a development switch)"""
    from rt_amd import abi
    shim = abi.load_shim()
    family = set()
    for k in range(shim.rt_hip_kernel_count()):
        family.add(shim.rt_hip_kernel_launches(k, None).decode())
    assert shim.rt_hip_kernel_launches(shim.rt_hip_kernel_count(), None) is None
    assert "pt_render_tiles_v0" not in family, "the literal kernel is a development-build member"
    seen = set()
    sizes = [dict(spheres=10), dict(spheres=126), dict(spheres=306), dict(spheres=5, meshes=1, tris=12),
             dict(spheres=8, meshes=1, tris=10240), dict(spheres=200, meshes=1, tris=100), dict(spheres=306, meshes=1, tris=100),
             dict(spheres=306, meshes=1, tris=600)]
    for integ, size, chk, refr, glass2, wide, rnd, depth, ws, wp in itertools.product(
            (PATH, CAST), sizes, (0, 1), (0, 1), (0, 1), (0, 1), (0, 1), (5, 30), (0, 1), (0, 1)):
        name = pick(integ, chk=chk, refr=refr, glass2=glass2, wide=wide, round_=rnd, depth=depth, park_ws=ws, wide_pend=wp, **size)
        assert name in family, (name, integ, size)
        seen.add(name)
    assert seen == family, f"rows no class reaches: {sorted(family - seen)}"
    assert len(family) == 35
