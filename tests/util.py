"""Shared helpers of the parity tests."""
import numpy as np

RMS_TOL = 1e-4  # BASELINE.json north_star: per-channel RMS of the linear float framebuffer

# how many oracle comparisons (assert_parity calls that passed) this process has made: tests/conftest.py attributes the
# kernel launches of a test to "under an oracle comparison" when this moved during it (tests/test_zz_kernel_coverage.py)
PARITY_PASSED = [0]


def channel_rms(a, b):
    d = np.asarray(a, dtype=np.float64).reshape(-1, 3) - np.asarray(b, dtype=np.float64).reshape(-1, 3)
    return np.sqrt((d * d).mean(axis=0))


def tile_pixels(width, height, tiles):
    """linear pixel indices (row-major inside each 8x8 tile) of the given tile ids, inside-image only"""
    tx = (width + 7) // 8
    out = []
    for t in tiles:
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        for r in range(8):
            for c in range(8):
                if x0 + c < width and y0 + r < height:
                    out.append((y0 + r) * width + x0 + c)
    return np.array(out, dtype=np.uint32)


def assert_parity(gpu_img, gpu_img8, gpu_stats, mean, rgb8, stats, what="", hdr=False, abs_floor=0.0):
    """gpu_img: (...,3) float32 linear; mean: same pixels, float64 oracle.
    The bar is the north star's ABSOLUTE 1e-4 per-channel RMS.  hdr=True (the fuzz scenes only, which
    ask for it explicitly) scales it with the largest value compared: a float32 framebuffer cannot
    hold 1e-4 absolute for the 1e5..1e6 radiances the reference's "fresnel" weights and HDR emitters
    produce there (6e-8 relative = 0.06 absolute at 1e6)."""
    g = np.asarray(gpu_img, dtype=np.float64).reshape(-1, 3)
    m = np.asarray(mean).reshape(-1, 3)
    assert g.shape == m.shape
    rms = channel_rms(g, m)
    rms_tol = RMS_TOL * (max(1.0, float(np.abs(m).max())) if hdr else 1.0)
    assert (rms <= rms_tol).all(), f"{what}: per-channel RMS {rms} > {rms_tol}"
    # much tighter than the north-star bar: only the float32 store (6e-8 relative) and the
    # fixed-point pixel sums of pt_render_tiles (absolute resolution <= (depth+2) * max
    # emission * spp * 2^-62, i.e. < 1e-13 on every configuration) may be visible
    err = np.abs(g - m)
    # (the fixed-point resolution is absolute and follows the brightest emitter -- a term must stay below 2^51 units --
    # so under hdr=True the floor scales with the largest value compared, like the RMS bar: 4e-12 for emitters of 1e3)
    # abs_floor: a caller that knows the scene may state the sums' resolution itself (fixed_point_floor: the brightest EMITTER
    # sets it, and it need not be in view -- tools/gpu_fuzz_parity.py scene 6035: 1.1e-12 off at a pixel value of 1.4e-8)
    floor = max(1e-12 * (max(1.0, float(np.abs(m).max())) if hdr else 1.0), abs_floor)
    bad = err > 1e-6 * np.abs(m) + floor
    assert not bad.any(), f"{what}: {bad.sum()} values off, worst {err[bad].max()} at value {np.abs(m)[bad][err[bad].argmax()]}"
    if gpu_img8 is not None:
        d8 = np.abs(np.asarray(gpu_img8, dtype=np.int16).reshape(-1, 3) - np.asarray(rgb8, dtype=np.int16).reshape(-1, 3))
        assert d8.max() <= 1, f"{what}: tonemapped bytes differ by {d8.max()} LSB"
    if gpu_stats is not None:
        # decision-exactness: every branch taken identically => identical counters
        assert gpu_stats["rays"] == stats["rays"], f"{what}: rays {gpu_stats['rays']} != {stats['rays']}"
        assert gpu_stats["tests"] == stats["tests"], f"{what}: tests {gpu_stats['tests']} != {stats['tests']}"
        if "casts" in stats:
            assert gpu_stats["casts"] == stats["casts"]
    PARITY_PASSED[0] += 1


def fixed_point_floor(sc):
    """what the pooled kernels' fixed-point pixel sums can resolve in a pixel MEAN of this scene: a term is an integer of
    scale 2^-k with (max_depth + 2) * max(emission, BACKGROUND) * 2^k < 2^51 (rt_hip_shim.hip: the launch's scale), rounded
    to nearest, and a sample has at most max_depth + 2 terms: (max_depth + 2)^2 * max emission * 2^-51"""
    emax = max([max(sc.objects[i].emission.x, sc.objects[i].emission.y, sc.objects[i].emission.z) for i in range(sc.n_objects)] +
               [max(sc.meshes[i].emission.x, sc.meshes[i].emission.y, sc.meshes[i].emission.z) for i in range(sc.n_meshes)] + [1.0])
    return (sc.max_depth + 2) ** 2 * emax * 2.0 ** -51


def untile_numpy(tiles, width, height, first, stride, count, image):
    """numpy statement of rt_hip_untile(): tiles [>=count, 64, 3] -> image [H, W, 3] in place"""
    tx = (width + 7) // 8
    for k in range(count):
        t = first + k * stride
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        block = np.asarray(tiles[k]).reshape(8, 8, 3)
        h = min(8, height - y0)
        w = min(8, width - x0)
        image[y0:y0 + h, x0:x0 + w] = block[:h, :w]
    return image


def glass_scene(width=96, height=64, samples=8, max_depth=5):
    """a small scene exercising every material branch of trace_path(): diffuse, mirror, light,
    M_REFRACTION (the reference's two-child 'glass', raytracer.c:514-529) and M_CHECKERED"""
    from rt_amd import abi, scene as S
    objs = [
        dict(flags=abi.M_DEFAULT | abi.M_CHECKERED, radius=10000.0, center=(0, -10005.0, 0), color=(0.8, 0.8, 0.8)),
        dict(flags=abi.M_DEFAULT, radius=4.0, center=(-11, -1, -2), color=(0.75, 0.25, 0.25)),
        dict(flags=abi.M_REFRACTION, radius=5.0, center=(0, 0, 0), color=(0.95, 0.95, 0.95)),
        dict(flags=abi.M_REFLECTION, radius=4.0, center=(11, -1, -3), color=(1, 1, 1)),
        dict(flags=abi.M_REFRACTION | abi.M_CHECKERED, radius=2.5, center=(5, -2.5, 8), color=(0.6, 0.9, 0.7)),
        dict(flags=abi.M_DEFAULT, radius=6.0, center=(-4, 18, 6), color=(1, 1, 1), emission=(5, 5, 5)),
    ]
    return S.custom_scene(objs, width, height, samples, max_depth, (0, 6, 38), (0, 0, 0))


def whitted_scene(width=96, height=64, samples=2, max_depth=5):
    """every branch of cast_ray() (raytracer.c:556-641): plain / checkered Phong surfaces, spheres
    around its fixed light at (2, 7, 2) so that shadow rays hit and miss, a mirror, a 'glass'
    sphere, and one with M_REFLECTION | M_REFRACTION (two children per hit)"""
    from rt_amd import abi, scene as S
    objs = [
        dict(flags=abi.M_DEFAULT | abi.M_CHECKERED, radius=10000.0, center=(0, -10005.0, 0), color=(0.8, 0.8, 0.8)),
        dict(flags=abi.M_DEFAULT, radius=4.0, center=(-11, -1, -2), color=(0.75, 0.25, 0.25)),
        dict(flags=abi.M_REFRACTION, radius=4.0, center=(-2, -1, 2), color=(0.95, 0.95, 0.95)),
        dict(flags=abi.M_REFLECTION, radius=4.0, center=(11, -1, -3), color=(1, 1, 1)),
        dict(flags=abi.M_REFLECTION | abi.M_REFRACTION, radius=2.5, center=(5, -2.5, 8), color=(0.6, 0.9, 0.7)),
        dict(flags=abi.M_DEFAULT | abi.M_CHECKERED, radius=2.0, center=(-6, -3, 9), color=(0.9, 0.8, 0.2)),
        dict(flags=abi.M_DEFAULT, radius=1.0, center=(2, 4, 2), color=(0.2, 0.3, 0.9)),
    ]
    return S.custom_scene(objs, width, height, samples, max_depth, (0, 6, 38), (0, 0, 0))


def mesh_soup_scene(seed=5, n_tris=40, width=64, height=48, samples=4, max_depth=8, checker=True, duplicates=True):
    """spheres + a random triangle soup with random texture coordinates, some triangles duplicated
    (exact ties between indices) and some degenerate (zero area, |a| < EPSILON): the scene on which
    the mesh scan of raytracer.c:417-435 is pinned.  With checker=True a checkered sphere stands in
    front of checkered triangles, which exercises the scan's stale hit.u / hit.v (ref_harness.c)."""
    from rt_amd import abi, scene as S
    rng = np.random.default_rng(seed)

    def tri(p0, p1, p2):
        return [tuple(p0) + tuple(rng.random(2)), tuple(p1) + tuple(rng.random(2)), tuple(p2) + tuple(rng.random(2))]
    tris = []
    for _ in range(n_tris):
        c = rng.uniform(-6, 6, 3)
        tris.append(tri(c + rng.normal(0, 2, 3), c + rng.normal(0, 2, 3), c + rng.normal(0, 2, 3)))
    if duplicates:
        tris += [tris[3], tris[7], tris[3]]                 # equal t at different indices: first index wins
        p = rng.uniform(-3, 3, 3)
        tris.append(tri(p, p, p + 1.0))                     # two equal vertices
        tris.append(tri(p, p + 1.0, p + 2.0))               # collinear
    chk = abi.M_CHECKERED if checker else 0
    objs = [dict(flags=abi.M_DEFAULT | chk, radius=3.0, center=(0, 0, 4), color=(.8, .7, .6)),
            dict(flags=abi.M_DEFAULT, radius=1e4, center=(0, -10008, 0), color=(.75, .75, .75)),
            dict(flags=abi.M_REFLECTION, radius=2.0, center=(-7, 0, 2), color=(1, 1, 1)),
            dict(flags=abi.M_DEFAULT, radius=2, center=(5, 9, 0), color=(1, 1, 1), emission=(5, 5, 5))]
    meshes = [dict(flags=abi.M_DEFAULT | chk, color=(.6, .8, .7), triangles=tris[:len(tris) // 2]),
              dict(flags=abi.M_DEFAULT, color=(.9, .5, .4), triangles=tris[len(tris) // 2:])]
    return S.custom_scene(objs, width, height, samples, max_depth, (0, 3, 30), (0, 0, 0), meshes=meshes)


def decode_png_rgb8(path):
    """minimal PNG reader for 8-bit RGB, non-interlaced, all five filter types -> (h, w, 3) uint8"""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        (n,), typ = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, interlace) == (8, 2, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    stride, bpp = w * 3, 3
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        f = raw[y * (stride + 1)]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=y * (stride + 1) + 1).astype(np.int32)
        cur = np.zeros(stride, dtype=np.int32)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0:
                pred = 0
            elif f == 1:
                pred = a
            elif f == 2:
                pred = b
            elif f == 3:
                pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, 3)


def convex_body_scene(seed, width=56, height=36, samples=4):
    """-> (scene, facets of the outer body): a random convex polyhedron (scipy's hull of 210 random points on an
    ellipsoid: ~400 facets of every shape), each facet wound at random, near or far from the origin, diffuse or mirror;
    on odd seeds a second body inside or beside it"""
    from scipy.spatial import ConvexHull
    from rt_amd import abi, scene as S
    rng = np.random.default_rng(1000 + seed)

    def body(n, radii, centre):
        p = rng.normal(size=(n, 3))
        p /= np.linalg.norm(p, axis=1)[:, None]
        p = p * np.asarray(radii) + np.asarray(centre)
        tris = []
        for a, b, c in ConvexHull(p).simplices:
            t = [tuple(p[a]), tuple(p[b]), tuple(p[c])]
            if rng.random() < 0.5:
                t = [t[0], t[2], t[1]]
            tris.append(t)
        return tris
    centres = [(0, 0, 0), (300, 100, -200), (0, 0, 0), (-40, 5, 10), (0, 0, 0), (2000, 0, 0), (0, 0, 0), (7, 1, 3)]
    centre = np.array(centres[seed % len(centres)], float)
    tris = body(210, rng.uniform(2, 5, 3), centre)
    meshes = [dict(flags=abi.M_REFLECTION if seed % 3 == 2 else abi.M_DEFAULT, color=(0.8, 0.6, 0.4), triangles=tris)]
    if seed % 2:
        off = np.array([0.5, 0.2, 0.1]) if seed % 4 == 1 else np.array([7.0, 0.5, 1.0])
        meshes.append(dict(flags=abi.M_DEFAULT, color=(0.3, 0.5, 0.9), triangles=body(40, rng.uniform(0.5, 1.5, 3), centre + off)))
    objs = [dict(flags=abi.M_DEFAULT, radius=1e4, center=tuple(centre + (0, -10008.0, 0)), color=(0.7, 0.7, 0.7)),
            dict(flags=abi.M_REFLECTION, radius=3.0, center=tuple(centre + (9, -1, 2)), color=(1, 1, 1)),
            dict(flags=abi.M_DEFAULT, radius=6.0, center=tuple(centre + (0, 24, 0)), color=(1, 1, 1), emission=(4, 4, 4))]
    cam = tuple(centre + rng.uniform(-1, 1, 3) * (6, 3, 6) + (0, 4, 22))
    return S.custom_scene(objs, width, height, samples, 8, cam, tuple(centre), meshes=meshes), len(tris)


def walls_scene(seed, width=72, height=44, samples=6, with_mesh=False):
    """a room of LEADING wall-sized spheres (what the kernels prune among themselves before the exact tests, BigPrune):
    2 to 8 walls of radius 1e3 .. 1e5 at random distances -- some pairs placed so that a camera ray meets both at (nearly)
    the same distance, exact duplicates included (index ties) --, mirror and diffuse, the camera sometimes almost on a
    wall or inside one; then lights and small spheres; optionally a mesh of > 256 triangles, which sends the scene
    to the parked-walk kernels (their sphere filter prunes too)"""
    from rt_amd import abi, scene as S
    rng = np.random.default_rng(7000 + seed)
    n_walls = [6, 2, 8, 4, 6, 3, 8, 6][seed % 8]
    objs = []
    axes = [(0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0), (0, 0, -1), (0, 0, 1), (0.6, 0.8, 0), (-0.6, 0, 0.8)]
    for k in range(n_walls):
        r = float(10.0 ** rng.uniform(3, 5))
        dist = float(rng.uniform(6, 30))
        ax = np.array(axes[k % len(axes)], float)
        c = ax * (r + dist)
        if k and rng.uniform() < 0.25:   # a second wall through (almost) the same points as the previous one
            prev = objs[-1]
            pc, pr = np.array(prev["center"]), prev["radius"]
            ax = pc / np.linalg.norm(pc)
            r = pr * float(rng.choice([1.0, 1.0 + 1e-9, 3.0]))
            c = ax * (np.linalg.norm(pc) - pr + r + float(rng.choice([0.0, 1e-7, 0.05, 0.3])))
        objs.append(dict(flags=int(rng.choice([abi.M_DEFAULT, abi.M_DEFAULT, abi.M_REFLECTION])), radius=r, center=tuple(c),
                         color=tuple(rng.uniform(0.3, 0.95, 3))))
    objs.append(dict(flags=abi.M_DEFAULT, radius=3.0, center=(0, 4, 0), color=(1, 1, 1), emission=(6, 5, 4)))
    for _ in range(int(rng.integers(2, 12))):
        objs.append(dict(flags=int(rng.choice([abi.M_DEFAULT, abi.M_REFLECTION])), radius=float(rng.uniform(0.3, 2.5)),
                         center=tuple(rng.uniform(-5, 5, 3)), color=tuple(rng.uniform(0.2, 1, 3))))
    cam = rng.uniform(-4, 4, 3)
    mode = seed % 4
    if mode == 1:    # a hair in front of the first wall
        w = objs[0]
        wc = np.array(w["center"])
        cam = wc - wc / np.linalg.norm(wc) * (w["radius"] + 1e-3)
    elif mode == 2:  # inside the first wall
        w = objs[0]
        wc = np.array(w["center"])
        cam = wc - wc / np.linalg.norm(wc) * (w["radius"] - 2.0)
    meshes = []
    if with_mesh:
        tris = []
        for _ in range(300):
            b = rng.uniform(-4, 4, 3)
            tris.append([tuple(b), tuple(b + rng.normal(size=3) * 0.7), tuple(b + rng.normal(size=3) * 0.7)])
        meshes = [dict(flags=abi.M_DEFAULT, color=(0.7, 0.8, 0.5), triangles=tris)]
    return S.custom_scene(objs, width, height, samples, 6, tuple(cam), tuple(rng.uniform(-1, 1, 3)), meshes=meshes)


def packed_room(n_packed, seed=1, width=1920, height=1080, samples=64, max_depth=16, glass=False):
    """config 4's room -- six wall-sized spheres, the ceiling light and the small magenta light, camera (0, 0, 50) --
    with `n_packed` spheres packed into it the way the reference's generate_random_spheres() (main.c:65-138) packs its 30:
    rejection sampling of non-overlapping spheres in the room's box with its material lottery (half emitters with a random
    colour, a fifth mirrors, a fifth glass, the rest white diffuse).  Radii are scaled by (30 / n)^(1/3) from its 2..8 so
    that any count fits; glass=False turns its M_REFRACTION spheres into diffuse ones (scenes with M_REFRACTION take
    the static kernels).  -> scene with 8 + n_packed spheres"""
    from rt_amd import abi, scene as S
    room = S.build_scene(4, width, height, samples)
    objs = []
    for i in range(8):   # walls and lights lead config 4's object array (host/scenes.c: build_room_walls, build_room_lights)
        o = room.objects[i]
        objs.append(dict(flags=int(o.flags), radius=float(o.radius), center=o.center.tuple(), color=o.color.tuple(),
                         emission=o.emission.tuple()))
    assert all(o["radius"] >= 1000 for o in objs[:6])
    rng = np.random.default_rng(4000 + seed)
    scale = min(1.0, (30.0 / max(n_packed, 1)) ** (1.0 / 3.0))
    aspect = width / height
    lo = np.array([-20.0 * aspect, -20.0, -30.0]) * 0.95
    hi = np.array([20.0 * aspect, 20.0, 30.0]) * 0.95
    cen = np.zeros((n_packed, 3))
    rad = np.zeros(n_packed)
    k = 0
    while k < n_packed:
        r = rng.uniform(2.0, 8.0) * scale
        c = rng.uniform(lo + r, hi - r)
        if k and (np.linalg.norm(cen[:k] - c, axis=1) < rad[:k] + r).any():
            continue
        cen[k], rad[k] = c, r
        k += 1
    for k in range(n_packed):
        u = rng.random()
        flags, color, emission = abi.M_DEFAULT, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)
        if u < 0.5:
            emission = tuple(rng.random(3))
        elif u > 0.8:
            flags = abi.M_REFRACTION if glass else abi.M_DEFAULT
        elif u > 0.6:
            flags = abi.M_REFLECTION
        objs.append(dict(flags=int(flags), radius=float(rad[k]), center=tuple(cen[k]), color=color, emission=emission))
    room.free()
    return S.custom_scene(objs, width, height, samples, max_depth, (0, 0, 50), (0, 0, 0))


def room_with_mesh(n_packed, seed, width, height, samples, max_depth, n_tris=600, checker=False):
    """packed_room(n_packed) -- more spheres than the LDS staging holds from ~250 on -- with a bumpy sheet of n_tris random
    triangles across it: the scene class of pt_render_tiles_tri_queued_mem[_chk] (checker: one packed sphere M_CHECKERED)"""
    from rt_amd import abi, scene as S
    room = packed_room(n_packed, seed, width, height, samples, max_depth)
    objs = [dict(flags=int(room.objects[i].flags) | (abi.M_CHECKERED if checker and i == 9 else 0), radius=float(room.objects[i].radius),
                 center=room.objects[i].center.tuple(), color=room.objects[i].color.tuple(), emission=room.objects[i].emission.tuple())
            for i in range(room.n_objects)]
    room.free()
    rng = np.random.default_rng(77 + seed)
    tris = []
    for _ in range(n_tris):
        c = np.array([rng.uniform(-25, 25), rng.uniform(-15, 15), rng.uniform(-20, 20)])
        a, b = rng.normal(size=3) * 1.5, rng.normal(size=3) * 1.5
        tris.append([tuple(c), tuple(c + a), tuple(c + b)])
    return S.custom_scene(objs, width, height, samples, max_depth, (0, 0, 50), (0, 0, 0),
                          meshes=[dict(flags=abi.M_DEFAULT, color=(0.8, 0.7, 0.6), triangles=tris)])


def fdlibm_atan2(y, x):
    """numpy statement of the kernels' atan2_tab (pt_math.h): fdlibm's e_atan2.c / s_atan.c with one division for all
    five reduction intervals, every operation unfused and in the same order -- so the device must agree BIT FOR BIT"""
    aT = [3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
          9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
          4.97687799461593236017e-02, -3.65315727442169155270e-02, 1.62858201153657823623e-02]
    hi = np.array([4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00])
    lo = np.array([2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17])
    pi, pi_lo = 3.1415926535897931160e+00, 1.2246467991473531772e-16
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    ay, ax = np.abs(y), np.abs(x)
    with np.errstate(all="ignore"):
        q = np.where(ay == 0.0, 0.0, ay / ax)
        idv = np.full(q.shape, 3)
        num = np.full(q.shape, -1.0)
        den = q.copy()
        for lim, k, n_, d_ in ((2.4375, 2, q - 1.5, 1.0 + 1.5 * q), (1.1875, 1, q - 1.0, q + 1.0), (0.6875, 0, 2.0 * q - 1.0, 2.0 + q),
                               (0.4375, -1, q, np.ones_like(q))):
            m = q < lim
            idv = np.where(m, k, idv)
            num = np.where(m, n_, num)
            den = np.where(m, d_, den)
        xr = num / den
        z = xr * xr
        w = z * z
        s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))))
        s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))))
        k = np.maximum(idv, 0)
        t = xr * (s1 + s2)
        r = np.where(idv < 0, xr - t, hi[k] - ((t - lo[k]) - xr))
        x_neg = np.signbit(x) & ((ax != 0.0) | (ay == 0.0))
        y_neg = np.signbit(y)
        left = np.where(y_neg, (r - pi_lo) - pi, pi - (r - pi_lo))
        right = np.where(y_neg, -r, r)
        return np.where(x_neg, left, right)


def lopsided_mesh_scene(seed):
    """three clusters of triangles whose sizes range over two decades, one holding most of them, plus exact duplicates and a
    degenerate triangle, under a light, on a floor, next to a mirror: where the two hierarchy builders differ most"""
    from rt_amd import abi, scene as S
    rng = np.random.default_rng(seed)
    tris = []
    for centre, spread, size, count in (((-6.0, 2.0, 0.0), 1.5, 0.05, 900), ((5.0, 3.0, -2.0), 4.0, 1.5, 60), ((0.0, 6.0, 4.0), 0.4, 0.01, 240)):
        for _ in range(count):
            c = np.asarray(centre) + rng.normal(size=3) * spread
            a, b = rng.normal(size=3) * size, rng.normal(size=3) * size
            tris.append([tuple(c), tuple(c + a), tuple(c + b)])
    tris += tris[:40]                                                  # exact duplicates: the (t, index) rule under both orders
    tris += [[(1.0, 1.0, 1.0), (1.0, 1.0, 1.0), (2.0, 1.0, 1.0)]]      # a degenerate one
    meshes = [dict(flags=abi.M_DEFAULT, color=(0.8, 0.7, 0.6), triangles=tris)]
    objs = [dict(flags=abi.M_DEFAULT, radius=4.0, center=(0, 18, 0), color=(1, 1, 1), emission=(8, 8, 8)),
            dict(flags=abi.M_DEFAULT, radius=1000.0, center=(0, -1004, 0), color=(0.6, 0.6, 0.6)),
            dict(flags=abi.M_REFLECTION, radius=2.0, center=(2, 0, 6), color=(0.9, 0.9, 0.9))]
    return S.custom_scene(objs, 64, 40, 6, 6, (4, 6, 22), (0, 2, 0), meshes=meshes)


def class_scene(n_packed=4, tris=0, chk=False, refr=False, glass2=False, wide=False, mesh_refr=False, mesh_chk=False,
                round_mesh=False, depth=5, width=48, height=32, samples=3, seed=1):
    """a small scene of a chosen CLASS of the kernel pick table (pt_kernel.hip, pt_pick_table): config 4's room with `n_packed`
    packed spheres (8 + n_packed spheres: <= 85 staged, 86..256 streamed by preference, beyond that too large to stage),
    optionally a mesh of `tris` triangles (a bumpy sheet; round_mesh: a tessellated ball instead, whose bounding sphere is the
    better probe), material flags on chosen objects, and -- wide -- a floor sphere of radius 1e19 through the room (a centre
    or radius beyond 1e17 is what `wide_range` means)"""
    from rt_amd import abi, scene as S
    room = packed_room(n_packed, seed, width, height, samples, depth)
    objs = [dict(flags=int(room.objects[i].flags), radius=float(room.objects[i].radius), center=room.objects[i].center.tuple(),
                 color=room.objects[i].color.tuple(), emission=room.objects[i].emission.tuple()) for i in range(room.n_objects)]
    room.free()
    if chk:
        objs[0]["flags"] |= abi.M_CHECKERED          # a wall
    if refr:
        objs[8]["flags"] = abi.M_REFRACTION          # the first packed sphere
        objs[8]["color"] = (0.95, 0.95, 0.95)
        objs[8]["emission"] = (0.0, 0.0, 0.0)
    if glass2:
        objs[9]["flags"] = abi.M_REFLECTION | abi.M_REFRACTION
        objs[9]["emission"] = (0.0, 0.0, 0.0)
    if wide:
        objs.append(dict(flags=abi.M_DEFAULT, radius=1e19, center=(0.0, -1e19 - 12.0, 0.0), color=(0.6, 0.7, 0.5)))
    meshes = []
    if tris:
        rng = np.random.default_rng(500 + seed)
        tl = []
        if round_mesh:
            n_lat = max(2, int(round((tris / 4.0) ** 0.5)))
            n_lon = max(3, tris // (2 * n_lat))
            c, r = np.array([2.0, -3.0, 8.0]), 7.0

            def p(i, j):
                th, ph = np.pi * i / n_lat, 2.0 * np.pi * j / n_lon
                return tuple(c + r * np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)]))
            for i in range(n_lat):
                for j in range(n_lon):
                    tl.append([p(i, j), p(i + 1, j), p(i + 1, j + 1)])
                    tl.append([p(i, j), p(i + 1, j + 1), p(i, j + 1)])
        else:
            for _ in range(tris):
                b = np.array([rng.uniform(-22, 22), rng.uniform(-14, 14), rng.uniform(-18, 18)])
                tl.append([tuple(b) + (0.0, 0.0), tuple(b + rng.normal(size=3) * 2.0) + (1.0, 0.0), tuple(b + rng.normal(size=3) * 2.0) + (0.0, 1.0)])
        flags = abi.M_REFRACTION if mesh_refr else abi.M_DEFAULT
        meshes = [dict(flags=flags | (abi.M_CHECKERED if mesh_chk else 0), color=(0.9, 0.85, 0.8), triangles=tl)]
    return S.custom_scene(objs, width, height, samples, depth, (0, 0, 50), (0, 0, 0), meshes=meshes)
