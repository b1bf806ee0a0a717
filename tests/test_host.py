"""Host-side logic of the boundary (no GPU): camera, primitives, OBJ loader, PNG writer,
scene builders, the shared RNG, tile partition arithmetic."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import SEED

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def host():
    from rt_amd import abi
    return abi.load_host()


@pytest.fixture(scope="module")
def prim():
    return np.load(os.path.join(GOLD, "primitives.npz"), allow_pickle=False)


def test_init_camera_matches_golden(prim):
    from rt_amd import scene as S
    for k in range(4):
        w, h = [int(v) for v in prim["cam_size"][k]]
        cam = S.make_camera(w, h, tuple(prim["cam_pose"][k][:3]), tuple(prim["cam_pose"][k][3:]))
        assert np.array_equal(np.frombuffer(bytes(cam), dtype=np.float64), prim["cam_frame"][k])


def test_host_primitives_match_golden(host, prim):
    from rt_amd import abi
    for k in range(0, len(prim["sph_ray"]), 3):
        r = prim["sph_ray"][k]
        ray = abi.Ray(abi.Vec3(*r[:3]), abi.Vec3(*r[3:]))
        hit = abi.Hit()
        ok = host.intersect_sphere(C.byref(ray), abi.Vec3(*prim["sph_center"][k]), float(prim["sph_radius"][k]),
                                   C.byref(hit))
        assert ok == bool(prim["sph_hit"][k])
        if ok:
            assert hit.t == prim["sph_t"][k]
    for k in range(0, len(prim["tri_ray"]), 3):
        r = prim["tri_ray"][k]
        ray = abi.Ray(abi.Vec3(*r[:3]), abi.Vec3(*r[3:]))
        v = prim["tri_verts"][k].reshape(3, 5)
        vs = [abi.Vertex(abi.Vec3(*row[:3]), abi.Vec2(*row[3:])) for row in v]
        hit = abi.Hit()
        ok = host.intersect_triangle(C.byref(ray), vs[0], vs[1], vs[2], C.byref(hit))
        assert ok == bool(prim["tri_hit"][k])
        if ok:
            assert [hit.t, hit.u, hit.v] == prim["tri_tuv"][k].tolist()
        n = host.calculate_surface_normal(vs[0].pos, vs[1].pos, vs[2].pos)
        assert list(n.tuple()) == prim["tri_normal"][k].tolist()


def test_counters_and_point_at(host):
    from rt_amd import abi
    before = C.c_longlong.in_dll(host, "intersection_test_count").value
    ray = abi.Ray(abi.Vec3(0, 0, 0), abi.Vec3(0, 0, 1))
    hit = abi.Hit()
    assert host.intersect_sphere(C.byref(ray), abi.Vec3(0, 0, 5), 1.0, C.byref(hit)) and hit.t == 4.0
    assert C.c_longlong.in_dll(host, "intersection_test_count").value == before + 1
    assert host.point_at(C.byref(ray), 2.5).tuple() == (0.0, 0.0, 2.5)
    assert host.clamp(abi.Vec3(-1, 0.5, 7)).tuple() == (0.0, 0.5, 1.0)


def test_host_rng_stream(host, pt):
    host.rt_set_seed(SEED)
    got = [host.random_double() for _ in range(8)]
    want = pt.random_doubles(SEED, 0xFFFFFFFF, 0xFFFFFFFF, 8)
    assert got == want.tolist()
    r = host.random_range(-3.0, 5.0)
    assert -3.0 <= r < 5.0


def test_rng_python_reimplementation(pt):
    """rt_rng.h restated in Python integers: the header is the single definition both the
    oracle and the kernel include; this guards it against accidental edits."""
    M = (1 << 64) - 1

    def mix(z):
        z ^= z >> 30
        z = (z * 0xBF58476D1CE4E5B9) & M
        z ^= z >> 27
        z = (z * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    def stream(seed, pixel, sample, n):
        h = mix((seed + 0x9E3779B97F4A7C15 * (pixel + 1)) & M)
        h = mix((h + 0xD1B54A32D192ED03 * (sample + 1)) & M) or 0x9E3779B97F4A7C15
        out = []
        for _ in range(n):
            h ^= (h << 13) & M
            h ^= h >> 7
            h ^= (h << 17) & M
            out.append((h >> 33) / 2147483648.0)
        return out
    for key in [(SEED, 0, 0), (SEED, 2073599, 1023), (2**63 + 5, 17, 4095)]:
        assert stream(*key, 12) == pt.random_doubles(*key, 12).tolist()


def test_load_obj_cube(host, pt):
    """load_obj() on an OBJ of config 3's cube (quads, v//vn faces, comments, mtl lines) == the
    procedural cube of scene 3; where the reference is mounted, its own assets/cube.obj too"""
    from rt_amd import abi, scene as S
    mesh = abi.TriangleMesh()
    assert host.load_obj(os.path.join(GOLD, "c3_cube.obj").encode(), C.byref(mesh))
    assert mesh.num_triangles == 12
    sc = S.build_scene(3, 64, 36, 1)
    assert sc.n_meshes == 1 and sc.meshes[0].mesh.num_triangles == 12
    host.rt_mesh_flip_winding(C.byref(mesh))
    for k in range(36):
        a, b = mesh.vertices[k], sc.meshes[0].mesh.vertices[k]
        # scene 3 = file vertices * 6 + (0, 1, 0)
        assert (a.pos.x * 6.0, a.pos.y * 6.0 + 1.0, a.pos.z * 6.0) == b.pos.tuple()
        assert (a.tex.x, a.tex.y) == (0.0, 0.0)
    # positions are floats widened to double (what the vendored OBJ parser yields)
    xs = sorted({mesh.vertices[k].pos.x for k in range(36)})
    assert float(np.float32(0.999999)) in xs and 0.999999 not in xs
    # outward normals after the flip, under the reference's winding formula
    for t in range(12):
        v = [mesh.vertices[3 * t + j].pos.tuple() for j in range(3)]
        n = pt.surface_normal(np.array(v).reshape(-1))
        assert np.dot(n, np.mean(v, axis=0)) > 0.5
    assert not host.load_obj(b"/nonexistent.obj", C.byref(mesh))
    ref_asset = "/root/reference/assets/cube.obj"  # build container only; nothing is copied
    if os.path.exists(ref_asset):
        other = abi.TriangleMesh()
        assert host.load_obj(ref_asset.encode(), C.byref(other)) and other.num_triangles == 12
        host.rt_mesh_flip_winding(C.byref(other))
        for k in range(36):
            assert other.vertices[k].pos.tuple() == mesh.vertices[k].pos.tuple()
    sc.free()


def test_load_obj_features(host, tmp_path):
    from rt_amd import abi
    p = tmp_path / "t.obj"
    p.write_text("# quad + tri, vt, negative indices\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                 "vn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4 -3 -2\n")
    mesh = abi.TriangleMesh()
    assert host.load_obj(str(p).encode(), C.byref(mesh))
    assert mesh.num_triangles == 3
    got = [(mesh.vertices[k].pos.tuple(), (mesh.vertices[k].tex.x, mesh.vertices[k].tex.y)) for k in range(9)]
    assert got[0] == ((0, 0, 0), (0, 0)) and got[1] == ((1, 0, 0), (1, 0)) and got[2] == ((1, 1, 0), (1, 1))
    assert got[3] == ((0, 0, 0), (0, 0)) and got[4] == ((1, 1, 0), (1, 1)) and got[5] == ((0, 1, 0), (0, 1))
    assert [g[0] for g in got[6:]] == [(0, 0, 0), (1, 0, 0), (1, 1, 0)] and got[6][1] == (0, 0)
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")
    assert not host.load_obj(str(bad).encode(), C.byref(mesh))


def test_png_writer_roundtrip(host, tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    path = str(tmp_path / "o.png")
    assert host.stbi_write_png(path.encode(), 17, 13, 3, img.ctypes.data, 17 * 3) == 1
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        (n,), typ = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(typ + body)
        chunks.append((typ, body))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (17, 13, 8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(13, 1 + 17 * 3)
    assert (raw[:, 0] == 0).all() and np.array_equal(raw[:, 1:].reshape(13, 17, 3), img)
    assert host.stbi_write_png(b"/nonexistent_dir/x.png", 17, 13, 3, img.ctypes.data, 51) == 0


def test_scene_builders(pt):
    from rt_amd import abi, scene as S
    counts = {1: (4, 0, 0), 2: (10, 0, 0), 3: (5, 1, 12), 4: (38, 0, 0), 5: (8, 1, 10240)}
    for cfg, (no, nm, nt) in counts.items():
        sc = S.build_scene(cfg, 64, 36, 1)
        assert (sc.n_objects, sc.n_meshes, sc.n_triangles) == (no, nm, nt)
        for i in range(no):
            o = sc.objects[i]
            assert o.flags in (abi.M_DEFAULT, abi.M_REFLECTION) and o.radius > 0
        sc.free()
    # config 2: spheres rest on the ground and do not overlap
    sc = S.build_scene(2)
    sph = [(np.array(sc.objects[i].center.tuple()), sc.objects[i].radius) for i in range(1, 10)]
    for i, (c, r) in enumerate(sph):
        assert abs(c[1] - (r - 5.0)) < 1e-12
        for c2, r2 in sph[i + 1:]:
            assert np.linalg.norm(c - c2) >= r + r2
    # config 4 follows the aspect ratio (reference main.c:244-247)
    a = S.build_scene(4, 1920, 1080, 1)
    assert a.objects[3].center.x == 10000 + 20 * (1920 / 1080)
    assert a.objects[36].emission.tuple() == (0.0, 0x32 * 15 / 255.0, 0xA0 * 15 / 255.0)
    # config 5: the UV sphere's non-degenerate triangles face outward under the reference's winding
    sc = S.build_scene(5, 64, 36, 1)
    m = sc.meshes[0].mesh
    centre = np.array([0.0, -8.0, 4.0])
    checked = 0
    for t in range(0, m.num_triangles, 97):
        v = np.array([m.vertices[3 * t + j].pos.tuple() for j in range(3)])
        if np.linalg.norm(np.cross(v[1] - v[0], v[2] - v[0])) < 1e-9:
            continue
        n = pt.surface_normal(v.reshape(-1))
        assert np.dot(n, v.mean(axis=0) - centre) > 0
        checked += 1
    assert checked > 50
    sc.free()


def test_tile_partition_arithmetic():
    from rt_amd import dist as D
    for (w, h) in [(1920, 1080), (37, 21), (8, 8), (9, 9), (3840, 2160)]:
        total = D.n_tiles(w, h)
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                f, s, c = D.rank_tiles(w, h, r, world)
                seen += [f + k * s for k in range(c)]
                assert c <= D.padded_count(w, h, world)
            assert sorted(seen) == list(range(total))


def test_sqrt_threshold_equivalence():
    """pt_trace.h (rejection_round) takes the square root of the rejection loop (reference raytracer.c:240,
    `while (vec3_length(p) > 1)`) out of the loop using  sqrt(x) > 1  <=>  x > 1 + 2^-52  for a
    correctly rounded sqrt.  Checked around the boundary and on random values."""
    import math
    eps = 2.0 ** -52
    thr = 1.0000000000000002
    assert thr == 1.0 + eps
    for k in range(-8, 64):
        x = 1.0 + k * eps if k >= 0 else 1.0 + k * eps / 2
        assert (math.sqrt(x) > 1.0) == (x > thr), (k, x)
    rng = np.random.default_rng(5)
    for x in np.concatenate([rng.uniform(0, 3, 20000), 1 + rng.uniform(-1e-12, 1e-12, 20000)]):
        assert (math.sqrt(float(x)) > 1.0) == (float(x) > thr)


def test_fused_range_mapping_is_exact():
    """rnd_pm1(): fma(r, 2^-30, -1) equals random_range(-1, 1) = (r / 2^31) * (1 - -1) + -1
    exactly, because r * 2^-30 - 1 is representable for every 31-bit r."""
    from fractions import Fraction
    rng = np.random.default_rng(6)
    for r in list(rng.integers(0, 2**31, 5000)) + [0, 1, 2**30, 2**30 + 1, 2**31 - 1]:
        r = int(r)
        ref = (r / 2147483648.0) * (1.0 - -1.0) + -1.0
        exact = Fraction(r, 2**30) - 1
        assert Fraction(ref) == exact  # no rounding anywhere: any evaluation order agrees


def test_division_by_small_integer_shortcut_is_exact():
    """div_small_int() in pt_scene_ctx.h: q0 = RN(a*y), r = fma(-q0, b, a), q = fma(r, y, q0) with
    y = RN(1/b) equals RN(a/b) for the numerators (x + r/2^31) and divisors (W-1, H-1) of
    raytracer.c:203-204.  Emulated here with exact rationals (float(Fraction) rounds correctly)."""
    from fractions import Fraction as Fr
    rng = np.random.default_rng(9)

    def fma(x, y, z):
        return float(Fr(x) * Fr(y) + Fr(z))

    divisors = [1, 2, 3, 7, 255, 599, 799, 1079, 1919, 2159, 3839, 4095, 65535, (1 << 20) - 1, (1 << 20)]
    for b in divisors:
        bf = float(b)
        y = 1.0 / bf
        xs = list(rng.integers(0, b + 1, 300)) + [0, 1, b - 1, b]
        for x in xs:
            for k in [0, 1, 2**31 - 1, 2**30, int(rng.integers(0, 2**31)), int(rng.integers(0, 2**31))]:
                a = float(x) + k / 2147483648.0
                q0 = a * y
                r = fma(-q0, bf, a)
                q = fma(r, y, q0)
                assert q == a / bf, (a, b)


def test_load_obj_survives_malformed_input_fuzz(host, tmp_path):
    """mutated, truncated, shuffled and token-soup OBJ files: load_obj() must either reject the
    file or return a consistent mesh -- never crash (tools/asan_host.sh runs 4000 of these under
    AddressSanitizer + UBSan)"""
    import random
    from rt_amd import abi
    libc = C.CDLL(None)
    rnd = random.Random(1)
    base = open(os.path.join(GOLD, "c3_cube.obj"), "rb").read()
    tokens = [b"v", b"vn", b"vt", b"f", b"o", b"g", b"s", b"usemtl", b"mtllib", b"#", b"1//1", b"1/2/3", b"-1", b"0",
              b"99999999999", b"1e400", b"nan", b"inf", b"-", b"/", b"//", b" ", b"\t", b"\r\n", b"\n", b"\x00", b"1.5",
              b"-2//-3", b"4/", b"/5"]
    loaded = 0
    path = str(tmp_path / "f.obj")
    for it in range(int(os.environ.get("RT_OBJ_FUZZ", "400"))):
        mode = it % 4
        if mode == 0:
            data = bytearray(base)
            for _ in range(rnd.randint(1, 8)):
                data[rnd.randrange(len(data))] = rnd.randrange(256)
        elif mode == 1:
            data = base[:rnd.randrange(len(base))]
        elif mode == 2:
            data = b"".join(rnd.choice(tokens) + rnd.choice([b" ", b"\n", b""]) for _ in range(rnd.randint(1, 200)))
        else:
            lines = base.split(b"\n")
            rnd.shuffle(lines)
            data = b"\n".join(lines * rnd.randint(1, 3))
        open(path, "wb").write(bytes(data))
        mesh = abi.TriangleMesh()
        if host.load_obj(path.encode(), C.byref(mesh)):
            loaded += 1
            for k in range(3 * mesh.num_triangles):      # every vertex must be readable
                _ = mesh.vertices[k].pos.x + mesh.vertices[k].tex.y
            libc.free(mesh.vertices)
    assert loaded > 0


def test_hierarchy_builder_invariants(tmp_path):
    """the host-side hierarchy builder (raytracer.c_amd/csrc/bvh_build.h: surface-area splits over 64 bins within the least depth
    leaves of PT_BVH_LEAF allow, median where no admissible split exists) compiled by g++ with the checker of
    tests/bvh_harness.cpp: order is a permutation, the leaves tile the triangle list, no leaf above PT_BVH_LEAF, the depth is the
    least possible (the kernels' LDS stacks are sized by it), every stored child box is exactly the union of its subtree, two
    builds agree -- on uniform, clustered, degenerate and huge inputs, under both builders"""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libbvh_harness.so")
    # (tools/asan_host.sh sets RT_BVH_HARNESS_FLAGS="-g -fsanitize=address,undefined" and preloads the sanitizer runtimes)
    # -DPT_DEV_KERNELS: the median-only arm (RT_HIP_BVH_MEDIAN) exists in development builds only; -Bsymbolic: the harness must call ITS
    # BvhBuild (inline members are weak symbols, and librt_hip.so -- loaded RTLD_GLOBAL by other tests -- exports the product's)
    subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-DPT_DEV_KERNELS", "-Wl,-Bsymbolic"] + os.environ.get("RT_BVH_HARNESS_FLAGS", "").split() + ["-I", os.path.join(root, "include"), "-I",
                    os.path.join(root, "raytracer.c_amd", "csrc"), os.path.join(root, "tests", "bvh_harness.cpp"), "-o", so], check=True)
    lib = C.CDLL(so)
    lib.bvh_check.restype = C.c_int
    lib.bvh_check.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32 * 5), C.POINTER(C.c_double)]

    def check(tris, median=False):
        g = np.ascontiguousarray(np.concatenate([tris[:, 0], tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]], axis=1), dtype=np.float64)
        out, cost = (C.c_uint32 * 5)(), C.c_double()
        if median:
            os.environ["RT_HIP_BVH_MEDIAN"] = "1"
        try:
            rc = lib.bvh_check(g.ctypes.data, len(g), C.byref(out), C.byref(cost))
        finally:
            os.environ.pop("RT_HIP_BVH_MEDIAN", None)
        assert rc == 0, f"invariant {rc} violated ({len(g)} triangles, median={median})"
        return dict(depth=out[0], max_depth=out[1], nodes=out[2], leaves=out[3], area=bool(out[4]), cost=cost.value)

    rng = np.random.default_rng(7)

    def soup(n, spread=10.0, size=0.3, centre=(0, 0, 0)):
        c = np.asarray(centre) + rng.normal(size=(n, 1, 3)) * spread
        return c + rng.normal(size=(n, 3, 3)) * size

    def uv_sphere(rings, segs, r=8.0):
        t = np.linspace(0, np.pi, rings + 1)[:, None]
        p = np.linspace(0, 2 * np.pi, segs + 1)[None, :]
        v = r * np.stack([np.sin(t) * np.cos(p), np.cos(t) * np.ones_like(p), np.sin(t) * np.sin(p)], axis=-1)
        a, b, c, d = v[:-1, :-1], v[1:, :-1], v[1:, 1:], v[:-1, 1:]
        return np.concatenate([np.stack([a, b, c], axis=-2).reshape(-1, 3, 3), np.stack([a, c, d], axis=-2).reshape(-1, 3, 3)])

    for n in (1, 2, 15, 16, 17, 30, 31, 240, 241, 255, 1000, 10240, 40000):       # leaf and depth boundaries (15 * 2^D)
        for median in (False, True):
            r = check(soup(n), median)
            least = 0
            while 15 * 2 ** least < n:
                least += 1
            assert r["max_depth"] == least and r["depth"] <= max(least, 1)
            assert r["leaves"] * 15 >= n and r["nodes"] == max(r["leaves"] - 1, 1)
    # lopsided: one dense cluster holding most of the triangles, two sparse ones with triangles 100x as large
    lop = np.concatenate([soup(3000, 0.5, 0.01, (-20, 0, 0)), soup(200, 6.0, 2.0, (15, 5, 0)), soup(40, 0.1, 0.5, (0, 30, 9))])
    # (little slack under the depth cap: 3,240 of 3,840 leaf slots -- greedy area splits near the root can use it up; the builder
    # builds both trees and keeps the cheaper by its own cost model, so it is never worse than the median tree)
    best, med = check(lop), check(lop, True)
    assert best["depth"] <= med["max_depth"] and best["cost"] <= med["cost"] and not med["area"]
    # config 5's kind of mesh: the area heuristic must pay within the median tree's depth
    sph = uv_sphere(64, 80)
    sah, med = check(sph), check(sph, True)
    assert len(sph) == 10240 and sah["depth"] == med["depth"] == 10 and sah["area"] and sah["cost"] < 0.92 * med["cost"]
    # degenerate inputs: all triangles identical, all centroids on one line / in one plane, zero-area triangles, huge and tiny
    # coordinates, duplicates -- the median fallback must take over wherever no area split exists
    one = np.tile(soup(1), (500, 1, 1))
    line = soup(700, 0.0, 0.2) + np.linspace(0, 50, 700)[:, None, None] * np.array([1.0, 0, 0])
    plane = soup(900) * np.array([1.0, 0.0, 1.0])
    points = np.repeat(rng.normal(size=(300, 1, 3)), 3, axis=1)
    for tris in (one, line, plane, points, soup(2000) * 1e12, soup(2000) * 1e-12, np.concatenate([soup(600)] * 3)):
        for median in (False, True):
            check(tris, median)
