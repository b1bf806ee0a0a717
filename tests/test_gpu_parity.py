"""GPU parity: the HIP path, called through the C-ABI (rt_hip.h), against the CPU oracle
(oracle/pt_oracle.c, itself pinned bit-for-bit to the compiled reference in
test_oracle_ref.py / test_golden.py).  Tolerance: BASELINE.json north_star, <= 1e-4
per-channel RMS on the linear float framebuffer; in addition the ray / test counters must
be EQUAL (every branch decision identical) and tonemapped bytes within 1 LSB.
"""
import numpy as np
import pytest

from conftest import ROOT, SEED
from util import assert_parity, tile_pixels

pytestmark = pytest.mark.gpu

# the development build of the shim (make shim-dev: -DPT_DEV_KERNELS): the switches that move a scene to another arm of the
# pick table, or swap the hierarchy builder, exist only there -- loaded by CHILD processes through RT_HIP_SHIM_PATH
DEV_LIB = __import__("os").path.join(ROOT, "raytracer.c_amd", "csrc", "librt_hip_dev.so")


@pytest.fixture(scope="module")
def gpu():
    import torch
    from rt_amd import abi, gpu as G
    assert abi.load_shim().rt_hip_device_count() >= 1, "no HIP device: the GPU tests must run on the GPU box"
    assert torch.cuda.is_available()
    return G


def _full(gpu, pt, sc, seed=SEED, integrator="path", hdr=False):
    gs = gpu.GpuScene(sc)
    img, img8, st = gs.render_image(seed, integrator=integrator)
    mean, rgb8, ost = pt.render_pixels(sc, seed, integrator=integrator)
    assert st["samples"] == sc.width * sc.height * sc.samples
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what=f"config {sc.config} {integrator}",
                  hdr=hdr)
    gs.close()
    return st


def test_config1_full(gpu, pt):
    """BASELINE configs[0]: 256x256, 4 spp, depth 4 -- whole image vs oracle."""
    from rt_amd import scene as S
    _full(gpu, pt, S.build_scene(1))


def test_config2_reduced(gpu, pt):
    from rt_amd import scene as S
    _full(gpu, pt, S.build_scene(2, 200, 152, 16))


def test_config4_reduced(gpu, pt):
    """the reference's own 38-sphere room (r = 1e4 wall spheres: fp32 would fail here)"""
    from rt_amd import scene as S
    _full(gpu, pt, S.build_scene(4, 160, 96, 16))


def test_config3_mesh_reduced(gpu, pt):
    from rt_amd import scene as S
    sc = S.build_scene(3, 240, 136, 8)
    _full(gpu, pt, sc)
    sc.free()


def test_ragged_size_and_odd_samples(gpu, pt):
    """width/height not multiples of the 8-pixel tile; spp not a multiple of the 4 slices"""
    from rt_amd import scene as S
    for (w, h, spp) in [(37, 21, 5), (9, 10, 1), (64, 8, 3)]:
        _full(gpu, pt, S.build_scene(1, w, h, spp))


def test_depth_zero_and_empty_scene(gpu, pt):
    from rt_amd import scene as S
    sc = S.build_scene(1, 32, 32, 4, max_depth=0)
    _full(gpu, pt, sc)
    empty = S.custom_scene([], 24, 16, 2, 4, (0, 0, 10), (0, 0, 0))
    st = _full(gpu, pt, empty)
    assert st["rays"] == 24 * 16 * 2 and st["tests"] == 0


def test_tile_partition_is_bit_invariant(gpu, pt):
    """rendering interleaved tile subsets (the multi-GPU partition) == one full render, bitwise"""
    import torch
    from rt_amd import scene as S
    sc = S.build_scene(2, 120, 72, 8)
    gs = gpu.GpuScene(sc)
    full, full8, st = gs.render_image(SEED)
    world = 3
    image = torch.zeros_like(full)
    image8 = torch.zeros_like(full8)
    tot = torch.zeros(4, dtype=torch.int64, device=full.device)
    for r in range(world):
        first, stride, count = gpu.rank_tiles(sc.width, sc.height, r, world)
        t, t8, s = gs.render_tiles(SEED, first, stride, count)
        gs.untile(t, t8, first, stride, count, image, image8)
        tot += s
    torch.cuda.synchronize()
    assert torch.equal(image, full) and torch.equal(image8, full8)
    assert tot.cpu().tolist()[0] == st["rays"]
    gs.close()


def test_run_to_run_determinism(gpu):
    import torch
    from rt_amd import scene as S
    sc = S.build_scene(4, 96, 56, 8)
    gs = gpu.GpuScene(sc)
    a, a8, sa = gs.render_image(SEED)
    b, b8, sb = gs.render_image(SEED)
    assert torch.equal(a, b) and torch.equal(a8, b8) and sa == sb
    c, _, _ = gs.render_image(SEED + 1)
    assert not torch.equal(a, c)
    gs.close()


def test_full_size_config4_tiles_vs_oracle(gpu, pt):
    """BASELINE configs[3] geometry at FULL 1920x1080 (reduced spp so the oracle finishes):
    whole-frame counters are self-consistent and 24 scattered tiles match the oracle."""
    from rt_amd import scene as S
    sc = S.build_scene(4, samples=8)
    gs = gpu.GpuScene(sc)
    img, img8, st = gs.render_image(SEED)
    assert st["samples"] == 1920 * 1080 * 8
    assert st["tests"] == st["casts"] * 38
    assert st["samples"] <= st["casts"] <= st["rays"] <= st["samples"] * (sc.max_depth + 2)
    rng = np.random.default_rng(7)
    tiles = rng.choice(gpu.n_tiles(1920, 1080), size=24, replace=False)
    px = tile_pixels(1920, 1080, tiles)
    mean, rgb8, _ = pt.render_pixels(sc, SEED, pixels=px)
    g = img.cpu().numpy().reshape(-1, 3)[px]
    g8 = img8.cpu().numpy().reshape(-1, 3)[px]
    assert_parity(g, g8, None, mean, rgb8, None, what="config 4 full-size tiles")
    gs.close()


def test_full_frame_1080p_every_pixel_vs_oracle(gpu, pt):
    """every one of the 2,073,600 pixels of the headline geometry (1 spp so that the
    single-threaded oracle finishes in seconds), and the whole frame's counters"""
    from rt_amd import scene as S
    sc = S.build_scene(4, samples=1)
    assert (sc.width, sc.height) == (1920, 1080)
    gs = gpu.GpuScene(sc)
    img, img8, st = gs.render_image(SEED)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what="config 4, full 1080p frame")
    gs.close()


def test_host_api_render_matches_tiles(gpu, pt):
    """render() of the raytracer.h boundary (C host -> rt_hip_render_image) == tile API"""
    import ctypes as C
    from rt_amd import abi, scene as S
    sc = S.build_scene(1, 64, 40, 4)
    host = abi.load_host()
    fb = np.zeros((sc.height, sc.width, 3), dtype=np.uint8)
    opt = abi.Options()
    opt.width, opt.height, opt.samples = sc.width, sc.height, sc.samples
    host.rt_set_max_depth(sc.max_depth)
    host.rt_set_seed(SEED)
    rc0 = C.c_longlong.in_dll(host, "ray_count").value
    host.render(fb.ctypes.data, sc.objects, sc.n_objects, C.byref(sc.camera), C.byref(opt))
    rays = C.c_longlong.in_dll(host, "ray_count").value - rc0
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert rays == ost["rays"]
    assert np.abs(fb.reshape(-1, 3).astype(np.int16) - rgb8.astype(np.int16)).max() <= 1


# ---- against the committed golden vectors (produced by the COMPILED REFERENCE) -----------

GOLD = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def test_golden_config1_frame(gpu):
    from rt_amd import scene as S
    fr = np.load(GOLD + "/frames.npz", allow_pickle=False)
    sc = S.build_scene(1, 64, 64, 4)
    gs = gpu.GpuScene(sc)
    img, img8, st = gs.render_image(SEED)
    ost = dict(rays=int(fr["c1_64_stats"][0]), tests=int(fr["c1_64_stats"][1]))
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, fr["c1_64_mean"], fr["c1_64_rgb8"], ost,
                  what="golden config 1 64x64")
    gs.close()


@pytest.mark.parametrize("tag,cfg", [("c1_s4", 1), ("c2_s64", 2), ("c4_s64", 4), ("c4_s1024", 4)])
def test_golden_full_size_tiles(gpu, tag, cfg):
    """tiles of the FULL-SIZE configurations (config 4 at its full 1024 spp included)"""
    import torch
    from rt_amd import scene as S
    fr = np.load(GOLD + "/frames.npz", allow_pickle=False)
    w, h, spp, depth = [int(v) for v in fr[tag + "_dims"]]
    sc = S.build_scene(cfg, w, h, spp)
    gs = gpu.GpuScene(sc)
    got, got8 = [], []
    stats = torch.zeros(4, dtype=torch.int64, device="cuda")
    for t in fr[tag + "_tiles"]:
        tl, tl8, _ = gs.render_tiles(SEED, int(t), 1, 1, stats=stats)
        got.append(tl[0])
        got8.append(tl8[0])
    torch.cuda.synchronize()
    g = torch.stack(got).cpu().numpy().reshape(-1, 3)
    g8 = torch.stack(got8).cpu().numpy().reshape(-1, 3)
    st = stats.cpu().tolist()
    ost = dict(rays=int(fr[tag + "_stats"][0]), tests=int(fr[tag + "_stats"][1]))
    assert_parity(g, g8, dict(rays=st[0], tests=st[2]), fr[tag + "_mean"], fr[tag + "_rgb8"], ost, what=tag)
    gs.close()


def test_untile_matches_numpy(gpu):
    import torch
    from rt_amd import scene as S
    from util import untile_numpy
    sc = S.build_scene(1, 37, 21, 1)
    gs = gpu.GpuScene(sc)
    total = gpu.n_tiles(37, 21)
    rng = np.random.default_rng(0)
    for first, stride in [(0, 1), (1, 3), (2, 5)]:
        count = (total - first + stride - 1) // stride
        t = rng.uniform(0, 1, (count, 64, 3)).astype(np.float32)
        t8 = rng.integers(0, 256, (count, 64, 3), dtype=np.uint8)
        img, img8 = gs.untile(torch.from_numpy(t).cuda(), torch.from_numpy(t8).cuda(), first, stride, count)
        torch.cuda.synchronize()
        want = untile_numpy(t, 37, 21, first, stride, count, np.zeros((21, 37, 3), dtype=np.float32))
        want8 = untile_numpy(t8, 37, 21, first, stride, count, np.zeros((21, 37, 3), dtype=np.uint8))
        assert np.array_equal(img.cpu().numpy(), want) and np.array_equal(img8.cpu().numpy(), want8)
    gs.close()


def test_render_image_c_host_entry(gpu, pt):
    """rt_hip_render_image(): host buffers in, synchronous -- what the C host calls"""
    from rt_amd import scene as S
    sc = S.build_scene(2, 72, 40, 8)
    img, img8, st, secs = gpu.render_image_host(sc, SEED, n_devices=1)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img, img8, st, mean, rgb8, ost, what="rt_hip_render_image")
    assert secs > 0


def test_golden_glass_frame(gpu):
    """device refraction tree (pending-ray stack) + checker texture vs the compiled reference"""
    from util import glass_scene
    fr = np.load(GOLD + "/frames.npz", allow_pickle=False)
    sc = glass_scene()
    gs = gpu.GpuScene(sc)
    img, img8, st = gs.render_image(SEED)
    ost = dict(rays=int(fr["glass_stats"][0]), tests=int(fr["glass_stats"][1]))
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, fr["glass_mean"], fr["glass_rgb8"], ost, what="glass")
    gs.close()


def test_refraction_depth_limit_is_reported(gpu):
    from rt_amd import gpu as G
    from util import glass_scene
    sc = glass_scene(32, 32, 1, max_depth=40)
    gs = gpu.GpuScene(sc)
    with pytest.raises(G.ShimError, match="max_depth <= 32"):
        gs.render_image(SEED)
    gs.close()


def test_cli_host_writes_the_image(gpu, pt, tmp_path):
    """the C command-line host (reference main.c flags -w -h -s -o) end to end: scene build,
    init_camera, render() on the GPU, PNG out"""
    import os
    import struct
    import subprocess
    import zlib
    from rt_amd import abi, scene as S
    exe = os.path.join(abi.PKG_DIR, "host", "raytracer")
    out = str(tmp_path / "cli.png")
    r = subprocess.run([exe, "-w", "64", "-h", "40", "-s", "4", "-o", out, "-c", "1", "-d", "4"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "cast " in r.stdout and "rendering took" in r.stdout and "done." in r.stdout
    data = open(out, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, dims = 8, b"", None
    while pos < len(data):
        (n,), typ = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            dims = struct.unpack(">II", body[:8])
        if typ == b"IDAT":
            idat += body
        pos += 12 + n
    assert dims == (64, 40)
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(40, 1 + 64 * 3)
    img = raw[:, 1:].reshape(-1, 3)
    sc = S.build_scene(1, 64, 40, 4)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert np.abs(img.astype(np.int16) - rgb8.astype(np.int16)).max() <= 1
    rays = int([ln for ln in r.stdout.splitlines() if ln.startswith("cast ")][0].split()[1])
    assert rays == ost["rays"]
    # no arguments: usage + failure, like the reference (main.c:189-193)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "Usage:" in r.stderr


def test_sample_chunks_are_bit_invariant(gpu, pt):
    """splitting every tile's samples over several workgroups (finer work units for
    multi-GPU shards) changes nothing: integer partial sums are exact"""
    import torch
    from rt_amd import scene as S
    sc = S.build_scene(4, 96, 56, 22)  # 22 spp: chunks of unequal size
    gs = gpu.GpuScene(sc)
    total = gpu.n_tiles(sc.width, sc.height)
    ref_t, ref_t8, ref_s = gs.render_tiles(SEED, 0, 1, total)
    torch.cuda.synchronize()
    for chunks in (2, 3, 7, 22):
        t, t8, s = gs.render_tiles(SEED, 0, 1, total, chunks=chunks)
        torch.cuda.synchronize()
        assert torch.equal(t, ref_t) and torch.equal(t8, ref_t8), chunks
        assert torch.equal(s, ref_s), (chunks, s.tolist(), ref_s.tolist())
    assert gs.suggest_chunks(total, 22) == 1 and gs.suggest_chunks(100, 1024) > 1
    with pytest.raises(gpu.ShimError):
        gs.render_tiles(SEED, 0, 1, total, chunks=23)
    gs.close()


def test_reference_main_runs_on_the_gpu_through_the_boundary(gpu, pt, tmp_path):
    """oracle/_ref/ref_main_dropin = the reference's main.c, unmodified, compiled against
    include/raytracer.h and linked with libraytracer_amd.so (oracle/Makefile).  Its own scene
    literal (main.c:256-397), its own init_camera()/render() calls and its own stb PNG writer,
    with our GPU path behind render(): the image must be the oracle's for that scene."""
    import os
    import subprocess
    from rt_amd import abi, scene as S
    from util import decode_png_rgb8
    exe = os.path.join(abi.REPO_ROOT, "oracle", "_ref", "ref_main_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_main_dropin not built (needs /root/reference at build time)")
    out = str(tmp_path / "ref_main.png")
    r = subprocess.run([exe, "-w", "96", "-h", "54", "-s", "8", "-o", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    img = decode_png_rgb8(out).reshape(-1, 3)
    sc = S.build_scene(4, 96, 54, 8, max_depth=5)  # the library defaults: MAX_DEPTH 5, seed 1666943821
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert np.abs(img.astype(np.int16) - rgb8.astype(np.int16)).max() <= 1
    rays = int([ln for ln in r.stdout.splitlines() if ln.startswith("cast ")][0].split()[1])
    tests = int([ln for ln in r.stdout.splitlines() if ln.startswith("checked ")][0].split()[1])
    assert (rays, tests) == (ost["rays"], ost["tests"])


def test_config5_mesh_hierarchy_reduced(gpu, pt):
    """BASELINE configs[4] scene (room + 10,240-triangle mesh) at reduced size: the triangles
    go through the bounding-volume hierarchy on the device; the oracle scans them linearly.
    Equal counters = every ray found the same closest primitive (incl. the tie rule)."""
    from rt_amd import scene as S
    sc = S.build_scene(5, 64, 36, 3)
    st = _full(gpu, pt, sc)
    assert st["tests"] == st["casts"] * (8 + 10240)
    sc.free()


def test_mesh_hierarchy_with_duplicate_triangles(gpu, pt):
    """exact ties: every triangle of a mesh duplicated (same t from two indices) -- the lower
    index must win under hierarchy order exactly as under the reference's linear order"""
    import math
    from rt_amd import abi, scene as S
    tris = []
    n = 12
    for i in range(n):
        for j in range(n):
            x0, x1 = -12 + 2 * i, -10 + 2 * i
            z0, z1 = -12 + 2 * j, -10 + 2 * j
            y = 1.0 + 0.5 * math.sin(i) * math.cos(j)
            a, b, c, d = (x0, y, z0), (x1, y, z0), (x1, y + 0.3, z1), (x0, y + 0.3, z1)
            tris += [[a, c, b], [a, d, c]]
    meshes = [dict(flags=abi.M_DEFAULT, color=(0.7, 0.6, 0.5), triangles=tris + tris),          # duplicated
              dict(flags=abi.M_REFLECTION, color=(1, 1, 1), triangles=[[(-30, -3, -30), (30, -3, 30), (30, -3, -30)],
                                                                          [(-30, -3, -30), (-30, -3, 30), (30, -3, 30)]])]
    objs = [dict(flags=abi.M_DEFAULT, radius=5.0, center=(0, 14, 0), color=(1, 1, 1), emission=(6, 6, 6))]
    sc = S.custom_scene(objs, 72, 40, 4, 6, (10, 14, 30), (0, 0, 0), meshes=meshes)
    assert sc.n_triangles == 4 * n * n + 2 > 256
    _full(gpu, pt, sc)


def test_device_math_shortcuts_are_bit_exact(gpu):
    """the kernel's exact-arithmetic shortcuts, evaluated on the device, against IEEE results on
    the host: sqrt without range scaling, division by a small integer through its reciprocal,
    the library sqrt / division themselves (correct rounding is what parity rests on)"""
    import ctypes as C
    from rt_amd import abi
    shim = abi.load_shim()
    rng = np.random.default_rng(12)

    def run(op, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros_like(a)
        rc = shim.rt_hip_selftest_math(op, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size, 0)
        assert rc == 0, shim.rt_hip_last_error()
        return out

    # sqrt: values shaped like r*r - d2 (cancellation leftovers), plain ranges, exact squares, zero
    r2 = rng.uniform(0.5, 9, 200000) ** 2
    x = np.concatenate([r2 - r2 * rng.uniform(0, 1, r2.size), 1e8 - rng.uniform(0, 1.2e6, 100000),
                        10.0 ** rng.uniform(-200, 200, 100000), rng.integers(1, 1 << 26, 50000).astype(np.float64) ** 2,
                        np.array([0.0, 1.0, 4.0, 2.0 ** -766, 2.0 ** -700, 1e-300 * 0 + 5e-217])])
    x = np.abs(x)
    one = np.ones_like(x)
    want = np.sqrt(x)
    assert np.array_equal(run(2, x, one), want), "library sqrt is not correctly rounded"
    assert np.array_equal(run(0, x, one), want), "unscaled sqrt differs from IEEE sqrt"
    # division by small integers (W-1, H-1) of numerators x + r / 2^31
    b = rng.choice([1, 2, 3, 255, 599, 799, 1079, 1919, 2159, 3839, 65535, (1 << 20) - 1], size=400000).astype(np.float64)
    a = np.floor(rng.uniform(0, 1, b.size) * (b + 1)) + rng.integers(0, 1 << 31, b.size) / 2147483648.0
    assert np.array_equal(run(3, a, b), a / b), "IEEE division differs"
    assert np.array_equal(run(1, a, b), a / b), "reciprocal shortcut differs from IEEE division"
    # fused [-1, 1) mapping
    r = rng.integers(0, 1 << 31, 200000).astype(np.float64)
    assert np.array_equal(run(4, r, r), (r / 2147483648.0) * (1.0 - -1.0) + -1.0)
    # reciprocal without range scaling: lengths of accepted rejection samples (sqrt of sums of squares
    # of multiples of 2^-30, <= 1), a broad range, and the ends of its stated domain
    q = (rng.integers(0, 1 << 31, (300000, 3)) / 1073741824.0) - 1.0
    l2 = (q * q).sum(axis=1)
    lens = np.sqrt(l2[(l2 <= 1.0000000000000002) & (l2 > 0)])
    y = np.concatenate([lens, 10.0 ** rng.uniform(-150, 150, 200000),
                        np.array([2.0 ** -30, 1.0, 1.0 - 2.0 ** -53, 2.0 ** -500, 2.0 ** 500, 3.0, 1.0 / 3.0])])
    assert np.array_equal(run(5, y, y), 1.0 / y), "unscaled reciprocal differs from IEEE division"
    # ... and of either sign: intersect_triangle's 1.0 / a (exact_triangle<.., UNSCALED>), |a| >= 1e-8
    z = np.concatenate([-y, 10.0 ** rng.uniform(-8, 36, 200000) * rng.choice([-1.0, 1.0], 200000), np.array([1e-8, -1e-8, -3.0])])
    assert np.array_equal(run(5, z, z), 1.0 / z), "unscaled reciprocal of a negative value differs from IEEE division"
    # atan2_tab (M_CHECKERED's texture coordinate, raytracer.c:410): bit for bit the numpy statement of the same fdlibm
    # algorithm, and within two ulps of the host's atan2 (glibc through numpy) -- on unit normals, which is what the kernels
    # pass, on wide-ranging arguments, and on the axis / zero / sign-of-zero cases
    from util import fdlibm_atan2
    nrm = rng.normal(size=(400000, 3))
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    ya = np.concatenate([nrm[:, 0], rng.normal(size=200000) * 10.0 ** rng.uniform(-12, 12, 200000),
                         np.array([0.0, -0.0, 1.0, -1.0, 0.0, -0.0, 0.0, -0.0, 1e-300, 1.0, 1.0, 0.4375, 0.6875, 1.1875, 2.4375, -2.4375])])
    xa = np.concatenate([nrm[:, 2], rng.normal(size=200000) * 10.0 ** rng.uniform(-12, 12, 200000),
                         np.array([1.0, 1.0, 0.0, -0.0, -1.0, -1.0, -0.0, 0.0, 1.0, 1e300, -1e300, 1.0, 1.0, 1.0, 1.0, -1.0])])
    got = run(6, ya, xa)
    assert np.array_equal(got.view(np.uint64), fdlibm_atan2(ya, xa).view(np.uint64)), "atan2_tab is not the fdlibm algorithm it states"
    want = np.arctan2(ya, xa)
    assert (np.abs(got - want) <= 2.0 * np.spacing(np.abs(want))).all() and np.array_equal(np.signbit(got), np.signbit(want))
    assert (got == want).mean() > 0.8
    # frac1(x) = fmod(x, 1.0), exactly: texture coordinates times the checker's 1e5 (or 10), either sign, and the edges
    fa = np.concatenate([rng.uniform(-1, 2, 300000) * 100000.0, rng.uniform(-1, 2, 100000) * 10.0, 10.0 ** rng.uniform(-300, 300, 100000),
                         np.array([0.0, -0.0, 0.5, 1.0, -1.0, 1.5, -1.5, 2.0 ** 52, 2.0 ** 52 + 1.0, 4503599627370495.5])])
    assert np.array_equal(run(7, fa, fa).view(np.uint64), np.fmod(fa, 1.0).view(np.uint64)), "frac1 differs from fmod(x, 1)"
    # win_add (the pooled refraction kernel's pixel sums): 300,000 values of either sign over 60 decades, added by 65,536 threads in
    # whatever order into ONE windowed accumulator -- the six words, combined as integers, must be EXACTLY the sum of the values'
    # truncations to multiples of 2^-64 (exact rational arithmetic on the host), and what win_add must refuse is counted
    from fractions import Fraction
    wv = np.concatenate([rng.normal(size=150000) * 10.0 ** rng.uniform(-30, 30, 150000), rng.uniform(-8, 8, 149990),
                         np.array([0.0, -0.0, 2.0 ** -64, -(2.0 ** -64), 2.0 ** -65, 2.0 ** 127, -(2.0 ** 127), 1.0, -1.0, 5e-324])])
    refuse = np.array([np.inf, -np.inf, np.nan, 2.0 ** 128, -(2.0 ** 130), 1e300])
    words = run(8, np.concatenate([wv, refuse]), np.zeros(len(wv) + len(refuse)))[:7].view(np.int64)
    assert words[6] == len(refuse)
    got = sum(int(words[k]) << (32 * k) for k in range(6))
    want = 0
    for x in wv.tolist():
        fr = Fraction(abs(x)) * (1 << 64)
        want += (fr.numerator // fr.denominator) * (-1 if np.signbit(x) else 1)
    assert got == want, "the windowed sum is not the exact sum of the truncated terms"


def _random_scene(seed, with_mesh, n_tris, extra_flags=(), materials="all"):
    """a deliberately nasty random scene: overlapping / nested / touching spheres, radii from
    1e-3 to 1e4, every material flag, HDR emission, camera possibly inside a sphere, optional
    triangle soup with degenerate and duplicated triangles.
    materials: "all" (M_REFRACTION included: the static `_refr` kernels render such scenes), "no_glass" (the pooled
    `_chk` / parked-walk `_chk` kernels), "plain" (diffuse / mirror only: the headline kernel family)"""
    from rt_amd import abi, scene as S
    rng = np.random.default_rng(1000 + seed)
    flags = [abi.M_DEFAULT, abi.M_REFLECTION, abi.M_REFRACTION, abi.M_DEFAULT | abi.M_CHECKERED,
             abi.M_REFLECTION | abi.M_CHECKERED] + list(extra_flags)
    if materials == "no_glass":
        flags = [f for f in flags if not f & abi.M_REFRACTION]
    elif materials == "plain":
        flags = [abi.M_DEFAULT, abi.M_REFLECTION]
    objs = []
    for k in range(int(rng.integers(1, 70))):
        r = float(10.0 ** rng.uniform(-3, 1.3)) if rng.uniform() < 0.9 else float(10.0 ** rng.uniform(2, 4))
        c = rng.uniform(-12, 12, 3)
        if r > 50:  # a "wall": push it away so that it bounds rather than swallows the scene
            axis = int(rng.integers(0, 3))
            c[axis] = np.sign(c[axis] or 1.0) * (r + rng.uniform(5, 30))
        if k and rng.uniform() < 0.15:  # touching / nested with the previous one
            c = np.array(objs[-1]["center"]) + rng.normal(size=3) * objs[-1]["radius"] * rng.choice([0.2, 1.0, 2.0])
        col = rng.uniform(0.05, 1.0, 3) if rng.uniform() < 0.8 else np.array([1.0, 1.0, 1.0])
        emi = rng.uniform(0, 1, 3) * float(10.0 ** rng.uniform(-1, 3)) if rng.uniform() < 0.25 else np.zeros(3)
        objs.append(dict(flags=int(rng.choice(flags)), radius=r, center=tuple(c), color=tuple(col), emission=tuple(emi)))
    meshes = []
    if with_mesh:
        tris = []
        for _ in range(n_tris):
            base = rng.uniform(-10, 10, 3)
            a, b, c = base, base + rng.normal(size=3) * 2, base + rng.normal(size=3) * 2
            kind = rng.uniform()
            if kind < 0.05:
                c = a + (b - a) * 0.5           # degenerate: zero area
            tri = [tuple(a) + tuple(rng.uniform(0, 1, 2)), tuple(b) + tuple(rng.uniform(0, 1, 2)),
                   tuple(c) + tuple(rng.uniform(0, 1, 2))]
            tris.append(tri)
            if kind > 0.9:
                tris.append(tri)                # exact duplicate: index tie
        half = len(tris) // 2
        meshes = [dict(flags=int(rng.choice(flags)), color=tuple(rng.uniform(0.2, 1, 3)), triangles=tris[:half]),
                  dict(flags=abi.M_DEFAULT, color=tuple(rng.uniform(0.2, 1, 3)),
                       emission=tuple(rng.uniform(0, 2, 3)), triangles=tris[half:])]
    cam = rng.uniform(-25, 25, 3)
    if rng.uniform() < 0.2 and objs:
        cam = np.array(objs[0]["center"]) + 0.3 * objs[0]["radius"]  # inside the first sphere
    if rng.uniform() < 0.15:
        cam = cam * 40.0                                              # far outside: exercises near_R
    w, h = int(rng.integers(9, 40)), int(rng.integers(9, 30))
    depth = int(rng.integers(0, 7))
    return S.custom_scene(objs, w, h, int(rng.integers(1, 7)), depth, tuple(cam), tuple(rng.uniform(-3, 3, 3)),
                          meshes=meshes)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_sphere_scenes(gpu, pt, seed):
    _full(gpu, pt, _random_scene(seed, False, 0), hdr=True)  # emitters up to 1e3, "fresnel" weights beyond 1


@pytest.mark.parametrize("seed,materials", [(s, m) for s in range(200, 208) for m in ("no_glass", "plain")])
def test_fuzz_sphere_scenes_pooled_kernels(gpu, pt, seed, materials):
    """the same nasty scenes without M_REFRACTION (which sends a scene to the static kernels): the pooled kernels --
    swap, primary trips, per-tile culling, wall pruning, integer pixel sums under emitters of up to 1e3"""
    sc = _random_scene(seed, False, 0, materials=materials)
    gs = gpu.GpuScene(sc)
    assert "_refr" not in gs.kernel_name()
    gs.close()
    _full(gpu, pt, sc, hdr=True)


@pytest.mark.parametrize("seed,n_tris,materials", [(s, n, m) for s, n, m in zip(range(220, 228), [5, 60, 200, 300, 450, 900, 2000, 40],
                                                                               ["plain", "no_glass"] * 4)])
def test_fuzz_mesh_scenes_pooled_and_parked_walk_kernels(gpu, pt, seed, n_tris, materials):
    """random meshes without M_REFRACTION: the small-mesh pooled kernels and the parked-walk kernels (plain and _chk)"""
    sc = _random_scene(seed, True, n_tris, materials=materials)
    gs = gpu.GpuScene(sc)
    assert "_refr" not in gs.kernel_name()
    gs.close()
    _full(gpu, pt, sc, hdr=True)


@pytest.mark.parametrize("seed,n_tris", [(s, n) for s, n in zip(range(16, 28), [3, 10, 40, 120, 250, 300, 400, 700, 1000, 60, 500, 2000])])
def test_fuzz_mesh_scenes(gpu, pt, seed, n_tris):
    """small meshes go through the flat filter, larger ones (> 256 primitives) through the hierarchy"""
    _full(gpu, pt, _random_scene(seed, True, n_tris), hdr=True)


def test_obj_file_through_render_ex(gpu, pt):
    """load_obj() -> MeshObject -> render_ex() of the raytracer.h boundary (C host, meshes and
    linear float output), against the oracle on the same mesh"""
    import ctypes as C
    from rt_amd import abi, scene as S
    host = abi.load_host()
    mesh = abi.TriangleMesh()
    assert host.load_obj((GOLD + "/c3_cube.obj").encode(), C.byref(mesh))
    host.rt_mesh_flip_winding(C.byref(mesh))
    for k in range(36):  # scale the unit cube up so it is visible
        v = mesh.vertices[k]
        v.pos = abi.Vec3(v.pos.x * 4, v.pos.y * 4 - 1, v.pos.z * 4)
    sc = S.build_scene(1, 80, 48, 6)          # config 1's spheres + the cube
    meshes = (abi.MeshObject * 1)()
    meshes[0].flags = abi.M_REFLECTION
    meshes[0].color = abi.Vec3(0.9, 0.8, 0.7)
    meshes[0].emission = abi.Vec3(0, 0, 0)
    meshes[0].mesh = mesh
    sc.meshes, sc.n_meshes, sc.n_triangles = meshes, 1, 12
    fb = np.zeros((sc.height, sc.width, 3), dtype=np.uint8)
    lin = np.zeros((sc.height, sc.width, 3), dtype=np.float32)
    opt = abi.Options()
    opt.width, opt.height, opt.samples = sc.width, sc.height, sc.samples
    host.rt_set_max_depth(sc.max_depth)
    host.rt_set_seed(SEED)
    host.render_ex(fb.ctypes.data, lin.ctypes.data, sc.objects, sc.n_objects, meshes, 1, C.byref(sc.camera), C.byref(opt))
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(lin, fb, None, mean, rgb8, None, what="render_ex + OBJ")
    assert host.rt_last_ray_bounces() == ost["casts"] and host.rt_last_render_seconds() > 0


def test_cancel_flag_returns_the_finished_part(gpu, pt):
    """rt_set_cancel_flag(): with the flag already raised a long render stops after its first
    slab; the finished tiles equal the full render's, the rest is zero"""
    import ctypes as C
    from rt_amd import abi, scene as S
    host = abi.load_host()
    sc = S.build_scene(4, 1920, 1080, 100)   # 2.07e8 pixel-samples: above the slab threshold
    opt = abi.Options()
    opt.width, opt.height, opt.samples = sc.width, sc.height, sc.samples
    host.rt_set_max_depth(4)
    host.rt_set_seed(SEED)
    full = np.zeros((sc.height, sc.width, 3), dtype=np.uint8)
    host.render(full.ctypes.data, sc.objects, sc.n_objects, C.byref(sc.camera), C.byref(opt))
    assert host.rt_last_render_cancelled() == 0
    flag = C.c_int(1)
    host.rt_set_cancel_flag(C.byref(flag))
    part = np.zeros_like(full)
    host.render(part.ctypes.data, sc.objects, sc.n_objects, C.byref(sc.camera), C.byref(opt))
    host.rt_set_cancel_flag(None)
    assert host.rt_last_render_cancelled() == 1
    done = part.reshape(-1, 3).any(axis=1)
    frac = done.mean()
    assert 0.15 < frac < 0.35, frac          # one slab of four
    assert np.array_equal(part[part.any(axis=2)], full[part.any(axis=2)])
    flag.value = 0
    host.rt_set_cancel_flag(C.byref(flag))   # registered but never raised: complete image
    again = np.zeros_like(full)
    host.render(again.ctypes.data, sc.objects, sc.n_objects, C.byref(sc.camera), C.byref(opt))
    host.rt_set_cancel_flag(None)
    assert host.rt_last_render_cancelled() == 0 and np.array_equal(again, full)


# ---- cast_ray: the Whitted integrator on the other side of render()'s `#if 1` ---------------

def test_whitted_golden_frame_every_branch(gpu):
    """device cast_ray vs the reference's compiled cast_ray() (raytracer.c:556-641): Phong,
    checker (M = 10), shadow rays, mirror, 'refraction', a sphere with both flags"""
    from util import whitted_scene
    fr = np.load(GOLD + "/whitted.npz", allow_pickle=False)
    sc = whitted_scene()
    gs = gpu.GpuScene(sc)
    img, img8, st = gs.render_image(SEED, integrator="whitted")
    ost = dict(rays=int(fr["scene_stats"][0]), tests=int(fr["scene_stats"][1]))
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, fr["scene_mean"], fr["scene_rgb8"], ost, what="whitted")
    # the switch is per call: the same scene object still path-traces
    img2, _, st2 = gs.render_image(SEED)
    assert st2["rays"] != st["rays"] and not np.array_equal(img2.cpu().numpy(), img.cpu().numpy())
    gs.close()


@pytest.mark.parametrize("tag,cfg", [("c2_s4", 2), ("c4_s4", 4)])
def test_whitted_golden_full_size_tiles(gpu, tag, cfg):
    import torch
    from rt_amd import scene as S
    fr = np.load(GOLD + "/whitted.npz", allow_pickle=False)
    w, h, spp, depth = [int(v) for v in fr[tag + "_dims"]]
    sc = S.build_scene(cfg, w, h, spp)
    gs = gpu.GpuScene(sc)
    got, got8 = [], []
    stats = torch.zeros(4, dtype=torch.int64, device="cuda")
    for t in fr[tag + "_tiles"]:
        tl, tl8, _ = gs.render_tiles(SEED, int(t), 1, 1, stats=stats, integrator="whitted")
        got.append(tl[0])
        got8.append(tl8[0])
    torch.cuda.synchronize()
    g = torch.stack(got).cpu().numpy().reshape(-1, 3)
    g8 = torch.stack(got8).cpu().numpy().reshape(-1, 3)
    st = stats.cpu().tolist()
    ost = dict(rays=int(fr[tag + "_stats"][0]), tests=int(fr[tag + "_stats"][1]))
    assert_parity(g, g8, dict(rays=st[0], tests=st[2]), fr[tag + "_mean"], fr[tag + "_rgb8"], ost, what="whitted " + tag)
    gs.close()


def test_whitted_configs_vs_oracle(gpu, pt):
    """spheres (LDS filter), a small mesh (flat filter) and the 10,240-triangle mesh (hierarchy)"""
    from rt_amd import scene as S
    for cfg, w, h, spp in [(1, 96, 96, 3), (3, 120, 68, 2), (4, 160, 90, 2), (5, 64, 36, 2)]:
        sc = S.build_scene(cfg, w, h, spp)
        st = _full(gpu, pt, sc, integrator="whitted")
        assert st["casts"] > st["rays"]  # shadow scans
        sc.free()


@pytest.mark.parametrize("seed", range(40, 52))
def test_whitted_fuzz(gpu, pt, seed):
    from rt_amd import abi
    both = (abi.M_REFLECTION | abi.M_REFRACTION, abi.M_REFLECTION | abi.M_REFRACTION | abi.M_CHECKERED,
            abi.M_REFRACTION | abi.M_CHECKERED)
    n_tris = [0, 0, 0, 0, 0, 0, 5, 40, 200, 300, 600, 1500][seed - 40]
    _full(gpu, pt, _random_scene(seed, n_tris > 0, n_tris, extra_flags=both), integrator="whitted", hdr=True)


def test_whitted_through_the_host_api_and_cli(gpu, pt, tmp_path):
    """rt_set_integrator(RT_CAST_RAY) + render(), and the CLI's -i 1"""
    import ctypes as C
    import os
    import subprocess
    from rt_amd import abi, scene as S
    from util import decode_png_rgb8
    host = abi.load_host()
    sc = S.build_scene(1, 72, 40, 2)
    mean, rgb8, ost = pt.render_pixels(sc, SEED, integrator="whitted")
    fb = np.zeros((sc.height, sc.width, 3), dtype=np.uint8)
    opt = abi.Options()
    opt.width, opt.height, opt.samples = sc.width, sc.height, sc.samples
    host.rt_set_max_depth(sc.max_depth)
    host.rt_set_seed(SEED)
    host.rt_set_integrator(abi.CAST_RAY)
    try:
        assert host.rt_get_integrator() == abi.CAST_RAY
        host.rt_set_integrator(7)  # ignored
        assert host.rt_get_integrator() == abi.CAST_RAY
        before = C.c_longlong.in_dll(host, "ray_count").value
        host.render(fb.ctypes.data, sc.objects, sc.n_objects, C.byref(sc.camera), C.byref(opt))
        assert C.c_longlong.in_dll(host, "ray_count").value - before == ost["rays"]
    finally:
        host.rt_set_integrator(abi.TRACE_PATH)
    assert np.abs(fb.reshape(-1, 3).astype(np.int16) - rgb8.astype(np.int16)).max() <= 1
    exe = os.path.join(abi.PKG_DIR, "host", "raytracer")
    out = str(tmp_path / "whitted.png")
    r = subprocess.run([exe, "-w", "72", "-h", "40", "-s", "2", "-o", out, "-c", "1", "-d", str(sc.max_depth), "-i", "1"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    img = decode_png_rgb8(out).reshape(-1, 3)
    assert np.abs(img.astype(np.int16) - rgb8.astype(np.int16)).max() <= 1
    r = subprocess.run([exe, "-w", "72", "-h", "40", "-s", "2", "-o", out, "-i", "2"], capture_output=True, text=True)
    assert r.returncode != 0 and "Usage:" in r.stderr


def test_whitted_depth_limit_only_for_two_child_materials(gpu):
    from rt_amd import abi, gpu as G, scene as S
    from util import glass_scene, whitted_scene
    deep = glass_scene(24, 16, 1, max_depth=40)      # M_REFRACTION alone: one child per hit, no limit
    gs = gpu.GpuScene(deep)
    gs.render_image(SEED, integrator="whitted")
    with pytest.raises(G.ShimError, match="max_depth <= 32"):
        gs.render_image(SEED)                         # the path tracer's two-child refraction
    gs.close()
    gs = gpu.GpuScene(whitted_scene(24, 16, 1, max_depth=40))  # has M_REFLECTION | M_REFRACTION
    with pytest.raises(G.ShimError, match="max_depth <= 32"):
        gs.render_image(SEED, integrator="whitted")
    gs.close()


def test_wide_range_scene_keeps_the_nan_safe_filter(gpu, pt):
    """centres / radii beyond 1e17 would overflow the fp32 sums of the sign-test filter: such
    scenes must take the compare-based scalar-table kernels (spheres alone, and with a mesh)"""
    from rt_amd import abi, scene as S
    objs = [
        dict(flags=abi.M_DEFAULT, radius=1e19, center=(0, -1e19 - 5.0, 0), color=(0.7, 0.7, 0.7)),
        dict(flags=abi.M_DEFAULT, radius=4.0, center=(-6, -1, 0), color=(0.75, 0.25, 0.25)),
        dict(flags=abi.M_REFLECTION, radius=4.0, center=(6, -1, 0), color=(1, 1, 1)),
        dict(flags=abi.M_DEFAULT, radius=3.0, center=(0, 9, -4), color=(1, 1, 1), emission=(6, 6, 6)),
        dict(flags=abi.M_DEFAULT, radius=1e30, center=(0, 0, 1e30 + 60.0), color=(0.3, 0.4, 0.8)),
    ]
    _full(gpu, pt, S.custom_scene(objs, 64, 40, 6, 5, (0, 4, 30), (0, 0, 0)))
    # (a SMALL sphere that far out is refused: it would stretch the filter's range, near_R)
    far = objs + [dict(flags=abi.M_DEFAULT, radius=1.0, center=(1e25, 0, 0), color=(0.5, 0.5, 0.5))]
    gs = gpu.GpuScene(S.custom_scene(far, 16, 16, 1, 2, (0, 4, 30), (0, 0, 0)))
    with pytest.raises(gpu.ShimError, match="finite bound"):
        gs.render_image(SEED)
    gs.close()
    tri = [[(-3, -4.9, 3, 0, 0), (3, -4.9, 3, 1, 0), (0, 2, 3, 0, 1)], [(-3, -4.9, 3, 0, 0), (0, 2, 3, 0, 1), (3, -4.9, 3, 1, 0)]]
    meshes = [dict(flags=abi.M_DEFAULT, color=(0.9, 0.8, 0.2), triangles=tri)]
    _full(gpu, pt, S.custom_scene(objs, 48, 32, 4, 4, (0, 4, 30), (0, 0, 0), meshes=meshes))
    # (a mesh vertex that far out is refused like the small sphere: triangles always count towards near_R -- which is also
    # what bounds |e1||e2| in the triangle test's unscaled reciprocal, exact_triangle<.., UNSCALED>)
    far_tri = [[(-3, -4.9, 3, 0, 0), (1e18, -4.9, 3, 1, 0), (0, 2, 3, 0, 1)]]
    gs = gpu.GpuScene(S.custom_scene(objs[1:4], 16, 16, 1, 2, (0, 4, 30), (0, 0, 0),
                                     meshes=[dict(flags=abi.M_DEFAULT, color=(0.9, 0.8, 0.2), triangles=far_tri)]))
    with pytest.raises(gpu.ShimError, match="finite bound"):
        gs.render_image(SEED)
    gs.close()
    # a NON-FINITE vertex is refused when the scene is created (round-4 advisor finding: fmax(reach, NaN) used to drop it, the
    # hierarchy builder then sorted NaN centroids); the reference has no meaning for such a triangle either -- every comparison
    # of its test is false
    for bad in (float("nan"), float("inf")):
        nan_tri = [[(-3, -4.9, 3, 0, 0), (bad, -4.9, 3, 1, 0), (0, 2, 3, 0, 1)]]
        with pytest.raises(gpu.ShimError, match="non-finite coordinate"):
            gpu.GpuScene(S.custom_scene(objs[1:4], 16, 16, 1, 2, (0, 4, 30), (0, 0, 0),
                                        meshes=[dict(flags=abi.M_DEFAULT, color=(0.9, 0.8, 0.2), triangles=nan_tri)]))


def test_bench_two_ranks_rehearsal(gpu):
    """bench.py's N > 1 control flow (torch.distributed.run, RANK/WORLD_SIZE, tile partition,
    gather, untile on rank 0, max/sum reductions, one JSON line from rank 0) with two ranks
    sharing this box's one GPU: gloo instead of RCCL (RCCL refuses two ranks on one device)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", "2", "--spp", "4", "--steps", "1", "--warmup", "1", "--cpu-tiles", "0", "--no-configs"]
    env = dict(os.environ, RT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29671", os.path.join(root, "bench.py"),
                          "--gpus", "2"] + common, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d2 = json.loads(lines[0])
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common,
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert d2["n_gpus"] == 2 and d1["n_gpus"] == 1 and d2["scaling"] == "strong"
    # the same frame: identical ray and cast counts whatever the partition
    assert d2["ray_count_per_step"] == d1["ray_count_per_step"]
    assert d2["ray_bounces_per_step"] == d1["ray_bounces_per_step"]
    assert "REHEARSAL" in d2["config"]["parallelism"]
    # the N > 1 diagnostics: ranks the backend connected, where a frame's time goes, per-rank kernel times
    assert d2["ranks_seen"] == 2 and set(d2["phase_ms"]) >= {"render", "gather", "untile"}
    assert len(d2["rank_kernel_ms"]["per_rank"]) == 2 and d2["rank_kernel_ms"]["min"] > 0
    # the frame gathered from the two ranks holds the bits of a one-GPU render (2,048 of its tiles, compared on rank 0)
    assert d2["assembly_check"]["bit_identical_to_one_gpu"] is True and d2["assembly_check"]["tiles"] == 2048, d2["assembly_check"]
    assert "ranks_seen" not in d1 and "assembly_check" not in d1 and d1["roofline"]["kernel"] == "pt_render_tiles"


def test_bench_host_path_and_config_array(gpu):
    """`bench.py --host-path`: one process through rt_hip_render_image() (the C host's path; with
    --gpus N > 1 it drives N devices and RCCL inside the shim -- one device here); and the
    per-configuration array of the default N = 1 run"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # RT_HIP_FORCE_COMM=1: the bench's own child builds an RCCL communicator and sends the tile buffers through the grouped
    # send / recv block (one device, to itself) -- so that bench.py --host-path, not only tests/rccl_child.py, has met a
    # communicator before the first 8-GPU run
    hp = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--host-path", "--gpus", "1", "--config", "2",
                         "--spp", "4", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, cwd=root,
                        env=dict(os.environ, RT_HIP_FORCE_COMM="1"))
    assert hp.returncode == 0, hp.stderr[-2000:]
    d = json.loads([ln for ln in hp.stdout.splitlines() if ln.startswith("{")][0])["host_path"]
    assert d["n_devices"] == 1 and len(d["call_ms"]) == 2 and d["ray_bounces_per_s"] > 0
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "1", "--steps", "1", "--warmup", "1",
                          "--cpu-tiles", "0", "--c5-spp", "1"], capture_output=True, text=True, timeout=600, cwd=root)
    assert run.returncode == 0, run.stderr[-2000:]
    d = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][0])
    got = {c["config"]: c for c in d["configs"]}
    assert set(got) == {2, 3, 4, 5} or set(got) == {2, 3, 5}
    assert got[3]["kernel"] == "pt_render_tiles_tri" and got[5]["kernel"] == "pt_render_tiles_tri_queued_sph"
    assert all("error" not in c and c["kernel_ms"] > 0 and c["ray_bounces_per_s"] > 0 for c in got.values())
    # round 5: the C host's whole call rides in the N = 1 line -- on one device, on 8 LOGICAL devices mapped onto this GPU (the
    # multi-device path, bit-equal), and as the command-line host's process with its phase clock
    hp, h8, cli = d["host_path"], d["host_path_logical8"], d["cli_host"]
    assert "error" not in hp and hp["n_devices"] == 1 and hp["kernel"] == "pt_render_tiles" and min(hp["call_ms"]) >= min(hp["kernel_ms"]) > 0
    assert set(hp["phase_ms"]) >= {"context", "launch_to_idle", "copy_out"} and hp["first_call_ms"] > min(hp["call_ms"])
    assert "error" not in h8 and h8["n_devices"] == 8 and h8["logical_devices_on_one_gpu"] and h8["equals_one_device_frame"] is True
    assert "error" not in cli and cli["process_wall_ms"] > 0 and cli["rays"] == d["ray_count_per_step"]
    assert set(cli["phase_ms"]) == {"HIP_runtime_start", "context", "render", "copy_out", "PNG"}
    # every entry says whether its committed PMC / PT_DIAG figures still describe the kernels (never a stale number)
    for c in got.values():
        if c.get("pmc_stale"):
            assert c["valu_busy"] is None and c["traffic"] is None and c["pmc_stale_reason"]
        elif "pmc_stale" in c:
            assert c["pmc_source"] is not None and c["valu_busy"] is not None
        assert c["frac_executed"] is None or 0 < c["frac_executed"] < 1


def test_bench_parity_block_against_the_compiled_reference(gpu):
    """the same-run parity block of bench.py: the CPU leg's pixels -- the reference's compiled
    trace_path() -- against the frame the timed steps produced, counters against a re-render of exactly those tiles"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "4", "--width", "480", "--height", "270",
                          "--spp", "16", "--steps", "2", "--warmup", "1", "--cpu-tiles", "256", "--cpu-workers", "4", "--no-configs"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-1500:])
    d = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    p = d["parity"]
    assert p["ok"] and p["counters_equal"] and p["timed_frame_equals_rerender"], p
    assert p["pixels"] == 256 * 64 and p["spp"] == 16 and max(p["rms"]) <= 1e-4 and p["u8_max_diff"] <= 1
    assert p["rays"]["gpu"] == p["rays"]["cpu"] > 0 and p["ray_bounces"]["gpu"] == p["ray_bounces"]["cpu"] > 0
    assert d["cpu_baseline"]["value"] > 0 and "error" not in d
    # a mesh configuration goes through the reference's revived mesh scan (RefMeshOracle)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "3", "--width", "240", "--height", "136",
                          "--spp", "8", "--steps", "1", "--warmup", "1", "--cpu-tiles", "64", "--cpu-workers", "4", "--no-configs"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-1500:])
    d = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["parity"]["ok"] and "mesh" in d["parity"]["oracle"], d["parity"]


def test_mesh_hierarchy_partitions_and_chunks_are_bit_invariant(gpu):
    """the kernels with postponed hierarchy walks and carried-over direction sampling: when a ray
    is walked or a direction drawn depends on what else the wave holds, the image must not --
    interleaved tile subsets and sample chunks reproduce the full render bit for bit"""
    import torch
    from rt_amd import scene as S
    sc = S.build_scene(5, 72, 40, 10)
    gs = gpu.GpuScene(sc)
    total = gpu.n_tiles(sc.width, sc.height)
    ref_t, ref_t8, ref_s = gs.render_tiles(SEED, 0, 1, total)
    torch.cuda.synchronize()
    for chunks in (2, 5):
        t, t8, s = gs.render_tiles(SEED, 0, 1, total, chunks=chunks)
        torch.cuda.synchronize()
        assert torch.equal(t, ref_t) and torch.equal(t8, ref_t8) and torch.equal(s, ref_s), chunks
    full, full8 = gs.untile(ref_t, ref_t8, 0, 1, total)
    image, image8 = torch.zeros_like(full), torch.zeros_like(full8)
    tot = torch.zeros(4, dtype=torch.int64, device=full.device)
    for r in range(3):
        first, stride, count = gpu.rank_tiles(sc.width, sc.height, r, 3)
        t, t8, s = gs.render_tiles(SEED, first, stride, count)
        gs.untile(t, t8, first, stride, count, image, image8)
        tot += s
    torch.cuda.synchronize()
    assert torch.equal(image, full) and torch.equal(image8, full8) and torch.equal(tot, ref_s)
    gs.close()
    sc.free()


def test_many_spheres_beyond_the_lds_staging_budget(gpu, pt):
    """600 spheres (57 KB of geometry + materials: more than a workgroup stages in LDS): the
    scalar-table kernels read geometry and materials from memory; with and without a mesh"""
    from rt_amd import abi, scene as S
    rng = np.random.default_rng(77)
    objs = [dict(flags=abi.M_DEFAULT, radius=10000.0, center=(0, -10004.0, 0), color=(0.7, 0.7, 0.7))]
    flags = [abi.M_DEFAULT, abi.M_DEFAULT, abi.M_REFLECTION, abi.M_DEFAULT | abi.M_CHECKERED]
    for k in range(599):
        c = rng.uniform(-20, 20, 3)
        c[1] = rng.uniform(-3, 12)
        emi = tuple(rng.uniform(0, 4, 3)) if k % 9 == 0 else (0, 0, 0)
        objs.append(dict(flags=int(flags[k % 4]), radius=float(rng.uniform(0.3, 1.2)), center=tuple(c),
                         color=tuple(rng.uniform(0.2, 1, 3)), emission=emi))
    sc = S.custom_scene(objs, 56, 32, 3, 5, (0, 6, 45), (0, 2, 0))
    st = _full(gpu, pt, sc)
    assert st["tests"] == st["casts"] * 600
    tri = [[(-6, -3.9, 6, 0, 0), (6, -3.9, 6, 1, 0), (0, 9, 6, 0, 1)]]
    meshes = [dict(flags=abi.M_REFLECTION, color=(0.9, 0.9, 0.9), triangles=tri)]
    _full(gpu, pt, S.custom_scene(objs, 40, 24, 2, 4, (0, 6, 45), (0, 2, 0), meshes=meshes))
    _full(gpu, pt, S.custom_scene(objs, 40, 24, 2, 3, (0, 6, 45), (0, 2, 0), meshes=meshes), integrator="whitted")


def test_rooms_of_more_than_256_spheres_take_the_pooled_body(gpu, pt):
    """The 257th sphere used to drop a scene onto the static in-memory kernel; now the pooled body runs with geometry
    and materials read from memory (pt_render_tiles_pool_mem*).  Rooms packed as the reference's
    generate_random_spheres() (main.c:65-138) would: on both sides of the staging budget, with sample chunks and a
    tile partition (bit-identical), and -- with its M_REFRACTION spheres kept -- the static kernel that such scenes
    still need"""
    import torch
    from util import packed_room
    sc = packed_room(60, 1, 72, 40, 6, 8)            # 68 spheres (6.5 KB of geometry + materials): staged in LDS
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name() == "pt_render_tiles"
    gs.close()
    _full(gpu, pt, sc)
    # beyond ~85 spheres a sphere-only scene is streamed although it would fit the staging budget (pt_prefer_streaming)
    for n, seed in ((100, 5), (248, 1), (249, 2), (700, 3)):
        sc = packed_room(n, seed, 72, 40, 6, 8)
        gs = gpu.GpuScene(sc)
        assert gs.kernel_name() == "pt_render_tiles_pool_mem_s", gs.kernel_name()
        total = gpu.n_tiles(sc.width, sc.height)
        ref_t, ref_t8, ref_s = gs.render_tiles(SEED, 0, 1, total)
        t, t8, st = gs.render_tiles(SEED, 0, 1, total, chunks=3)
        torch.cuda.synchronize()
        assert torch.equal(t, ref_t) and torch.equal(t8, ref_t8) and torch.equal(st, ref_s)
        first, stride, count = gpu.rank_tiles(sc.width, sc.height, 1, 3)
        p, p8, _ = gs.render_tiles(SEED, first, stride, count)
        torch.cuda.synchronize()
        assert torch.equal(p, ref_t[first::stride][:count]) and torch.equal(p8, ref_t8[first::stride][:count])
        gs.close()
        st = _full(gpu, pt, sc)
        assert st["tests"] == st["casts"] * (n + 8)
    # with the generator's glass (a fifth of the spheres M_REFRACTION): the pooled refraction kernel, streamed
    for n, seed in ((120, 6), (300, 4)):
        sc = packed_room(n, seed, 56, 32, 4, 6, glass=True)
        gs = gpu.GpuScene(sc)
        assert gs.kernel_name() == "pt_render_tiles_refr_pool_mem", gs.kernel_name()
        total = gpu.n_tiles(sc.width, sc.height)
        a_t, a_t8, a_s = gs.render_tiles(SEED, 0, 1, total)
        b_t, b_t8, b_s = gs.render_tiles(SEED, 0, 1, total)
        torch.cuda.synchronize()
        assert torch.equal(a_t, b_t) and torch.equal(a_t8, b_t8) and torch.equal(a_s, b_s)
        gs.close()
        _full(gpu, pt, sc, hdr=True)
    # ... and with a mesh next to the glass: the general static in-memory kernel
    from rt_amd import scene as S
    room = packed_room(300, 4, 40, 24, 2, 5, glass=True)
    objs = [dict(flags=int(room.objects[i].flags), radius=float(room.objects[i].radius), center=room.objects[i].center.tuple(),
                 color=room.objects[i].color.tuple(), emission=room.objects[i].emission.tuple()) for i in range(room.n_objects)]
    tri = [[(-6, -3.9, 6, 0, 0), (6, -3.9, 6, 1, 0), (0, 9, 6, 0, 1)]]
    sc = S.custom_scene(objs, 40, 24, 2, 5, (0, 0, 50), (0, 0, 0), meshes=[dict(flags=4, color=(0.9, 0.9, 0.9), triangles=tri)])
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name() == "pt_render_tiles_mem"
    gs.close()
    _full(gpu, pt, sc, hdr=True)


@pytest.mark.parametrize("checker", [False, True])
def test_rooms_beyond_the_staging_budget_with_a_large_mesh_take_parked_walks(gpu, pt, checker):
    """300 packed spheres (beyond the LDS staging) next to a mesh of 600 triangles: the parked-walk body with sphere geometry,
    materials and the spheres' filter pairs from memory (pt_render_tiles_tri_queued_mem[_chk], end of round 4; until then the
    lane-waiting pt_render_tiles_pool_mem_tri) -- frame and counters = oracle, partitions bit-equal"""
    import torch
    from util import room_with_mesh
    sc = room_with_mesh(300, 9, 56, 32, 4, 6, checker=checker)
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name() == ("pt_render_tiles_tri_queued_mem_chk" if checker else "pt_render_tiles_tri_queued_mem"), gs.kernel_name()
    img, img8, st = gs.render_image(SEED)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what="room of 308 spheres + 600 triangles")
    total = gpu.n_tiles(sc.width, sc.height)
    t_all, t8_all, _ = gs.render_tiles(SEED, 0, 1, total)
    t_odd, t8_odd, _ = gs.render_tiles(SEED, 1, 2, total // 2)
    torch.cuda.synchronize()
    assert torch.equal(t_all[1::2][: total // 2], t_odd[: total // 2]) and torch.equal(t8_all[1::2][: total // 2], t8_odd[: total // 2])
    gs.close()


def test_many_samples_per_pixel(gpu, pt):
    """16x16 pixels x 20,000 spp: long job pools (1,250 refills of a wave's sample queue), 5 million
    fixed-point additions per tile, and the same again split over 7 sample chunks"""
    import torch
    from rt_amd import scene as S
    sc = S.build_scene(1, 16, 16, 20000)
    gs = gpu.GpuScene(sc)
    img, img8, st = gs.render_image(SEED)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what="20000 spp")
    t, t8, _ = gs.render_tiles(SEED, 0, 1, 4, chunks=7)
    full, full8 = gs.untile(t, t8, 0, 1, 4)
    torch.cuda.synchronize()
    assert torch.equal(full, img) and torch.equal(full8, img8)
    gs.close()


# ---- triangle meshes against the compiled reference (tests/golden/meshes.npz: the reference's
# ---- compiled trace_path() + its commented-out mesh scan revived around its compiled primitives)

@pytest.mark.parametrize("tag,cfg", [("c3_s256", 3), ("c5_s4096", 5)])
def test_golden_mesh_tiles_full_size(gpu, tag, cfg):
    """BASELINE configs[2] at its full 1920x1080 x 256 spp (flat-filter triangle kernel) and configs[4]
    at its full 3840x2160 x 4096 spp (10,240 triangles: hierarchy kernel, W-1 = 3839 camera quotient,
    16 x 4096-job pools): each golden tile rendered on its own through render_tiles(first=t, count=1)"""
    import torch
    from rt_amd import scene as S
    fr = np.load(GOLD + "/meshes.npz", allow_pickle=False)
    w, h, spp, depth = [int(v) for v in fr[tag + "_dims"]]
    sc = S.build_scene(cfg)
    assert (sc.width, sc.height, sc.samples, sc.max_depth) == (w, h, spp, depth)
    gs = gpu.GpuScene(sc)
    got, got8 = [], []
    stats = torch.zeros(4, dtype=torch.int64, device="cuda")
    for t in fr[tag + "_tiles"]:
        tl, tl8, _ = gs.render_tiles(SEED, int(t), 1, 1, stats=stats)
        got.append(tl[0])
        got8.append(tl8[0])
    torch.cuda.synchronize()
    g = torch.stack(got).cpu().numpy().reshape(-1, 3)
    g8 = torch.stack(got8).cpu().numpy().reshape(-1, 3)
    st = stats.cpu().tolist()
    ost = dict(rays=int(fr[tag + "_stats"][0]), tests=int(fr[tag + "_stats"][1]))
    assert_parity(g, g8, dict(rays=st[0], tests=st[2]), fr[tag + "_mean"], fr[tag + "_rgb8"], ost, what=tag)
    gs.close()
    sc.free()


@pytest.mark.parametrize("integ", ["path", "whitted"])
def test_golden_mesh_soup_frames(gpu, integ):
    """textured + checkered triangle soup with duplicated and degenerate triangles, two meshes, under
    both integrators: the device reproduces the literal scan's hit.u / hit.v (those of the LAST
    triangle the ray passes, TriLast in pt_intersect.h) -- frames from the compiled reference"""
    from util import mesh_soup_scene
    fr = np.load(GOLD + "/meshes.npz", allow_pickle=False)
    sc = mesh_soup_scene()
    gs = gpu.GpuScene(sc)
    img, img8, st = gs.render_image(SEED, integrator=integ)
    ost = dict(rays=int(fr[f"soup_{integ}_stats"][0]), tests=int(fr[f"soup_{integ}_stats"][1]))
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, fr[f"soup_{integ}_mean"], fr[f"soup_{integ}_rgb8"], ost,
                  what=f"mesh soup {integ}")
    gs.close()


@pytest.mark.parametrize("integ", ["path", "whitted"])
def test_checkered_mesh_through_the_hierarchy(gpu, pt, integ):
    """the same rule in the hierarchy kernels (> 256 primitives): with M_CHECKERED materials and
    triangles the walk may not prune by the closest hit, every passing triangle matters"""
    from util import mesh_soup_scene
    sc = mesh_soup_scene(seed=13, n_tris=420, width=56, height=40, samples=3, max_depth=5)
    assert sc.n_objects + sc.n_triangles > 256
    _full(gpu, pt, sc, integrator=integ)


# ---- the reference's primitive known-answers on the DEVICE (tests/golden/primitives.npz) ----

def _selftest_intersect(kind, rays, prims, near_R):
    import ctypes as C
    from rt_amd import abi
    shim = abi.load_shim()
    n = len(rays)
    rays = np.ascontiguousarray(rays, dtype=np.float64)
    prims = np.ascontiguousarray(prims, dtype=np.float64)
    hit = np.zeros(n, dtype=np.uint8)
    tuv = np.zeros((n, 3))
    keep = np.zeros((n, 3), dtype=np.uint64)
    rc = shim.rt_hip_selftest_intersect(kind, rays.ctypes.data, prims.ctypes.data, n, C.c_double(near_R),
                                        hit.ctypes.data, tuv.ctypes.data, keep.ctypes.data, 0)
    assert rc == 0, shim.rt_hip_last_error()
    return hit, tuv, keep


@pytest.mark.parametrize("near_R", [64.0, 3.0e4])
def test_device_sphere_known_answers(gpu, pt, near_R):
    """1,100 (ray, sphere) cases generated by the reference's compiled intersect_sphere() -- grazing,
    origin inside / on the surface, t ~ EPSILON, radius-1e4 walls (raytracer.c:77-118) -- through the
    kernel's exact_sphere: hit flags and t BIT-equal.  And the conservative phase-1 filter, in all
    three forms, over all 64 x 1,100 (ray, sphere) pairs of the blocks: it never drops a pair the
    exact test accepts."""
    pr = np.load(GOLD + "/primitives.npz", allow_pickle=False)
    rays, cen, rad = pr["sph_ray"], pr["sph_center"], pr["sph_radius"]
    n = len(rays)
    hit, tuv, keep = _selftest_intersect(0, rays, np.concatenate([cen, rad[:, None]], axis=1), near_R)
    assert np.array_equal(hit, pr["sph_hit"])
    on = hit.astype(bool)
    assert np.array_equal(tuv[on, 0], pr["sph_t"][on]), "t of accepted sphere hits differs from the compiled reference"
    assert (tuv[~on, 0] == np.finfo(np.float64).max).all()
    inside = (rays[:, :3] ** 2).sum(axis=1) <= near_R * near_R
    dropped_total = 0
    for i in range(n):
        b = (i // 64) * 64
        for j in range(b, min(b + 64, n)):
            ok, _ = pt.intersect_sphere(rays[i], cen[j], rad[j])
            for f in range(3):
                kept = (int(keep[i, f]) >> (j - b)) & 1
                assert kept or not ok, f"filter form {f} dropped sphere {j} for ray {i}, which the exact test accepts"
                dropped_total += (not kept) and f == 0
            if not inside[i]:
                assert all((int(keep[i, f]) >> (j - b)) & 1 for f in range(3))  # far origins skip the filter
    # (a usefulness check, not a correctness one.  With near_R = 3e4 around unit-sized cases the thresholds' widening,
    # ~ near_R^2 * 2.4e-6, exceeds most squared distances: the sign-test form then keeps spheres BEHIND a ray whose
    # origin lies inside their widened radius -- it judges by the one sign of tca |tca| - ll since round 3 -- so less
    # is dropped there; real scenes have near_R of the order of their size)
    assert dropped_total > (0.25 if near_R < 1000 else 0.2) * n * 64, "the filter should reject most non-hitting pairs"


@pytest.mark.parametrize("near_R", [64.0, 3.0e4])
def test_device_triangle_known_answers(gpu, pt, near_R):
    """700 (ray, triangle) cases from the reference's compiled intersect_triangle() -- aimed at edges /
    vertices, parallel to the plane (raytracer.c:120-174) -- through the kernel's exact_triangle: hit
    flags, t and the texture coordinates blended from its barycentrics BIT-equal; the filter (bounding
    spheres) never drops an accepted pair."""
    pr = np.load(GOLD + "/primitives.npz", allow_pickle=False)
    rays, verts = pr["tri_ray"], pr["tri_verts"].reshape(-1, 3, 5)
    n = len(rays)
    pos = verts[:, :, :3].reshape(n, 9)
    hit, tuv, keep = _selftest_intersect(1, rays, pos, near_R)
    assert np.array_equal(hit, pr["tri_hit"])
    on = hit.astype(bool)
    assert np.array_equal(tuv[on, 0], pr["tri_tuv"][on, 0])
    # raytracer.c:154-167: tex = (st0 * (1 - u - v) + st1 * u) + st2 * v, as the render kernels blend it
    u, v = tuv[:, 1], tuv[:, 2]
    st = verts[:, :, 3:]
    w0 = 1 - u - v
    tex = (st[:, 0] * w0[:, None] + st[:, 1] * u[:, None]) + st[:, 2] * v[:, None]
    assert np.array_equal(tex[on], pr["tri_tuv"][on, 1:])
    n_kept = n_pre = n_pairs = 0
    for i in range(n):
        b = (i // 64) * 64
        for j in range(b, min(b + 64, n)):
            ok, _ = pt.intersect_triangle(rays[i], pr["tri_verts"][j])
            n_pairs += 1
            for f in (0, 1, 2):  # 0: the per-lane fp32 Moeller-Trumbore pre-test; 1, 2: the bounding-sphere filter
                kept = (int(keep[i, f]) >> (j - b)) & 1
                assert kept or not ok, f"filter form {f} dropped triangle {j} for ray {i}, which the exact test accepts"
                n_kept += kept and f > 0
                n_pre += kept and f == 0
    if near_R < 100:  # (at near_R = 3e4 the d2 tolerance, 32 e near_R^2 ~ 1.7e3, lets nearly every pair through)
        assert n_kept < 0.8 * 2 * n_pairs
        assert n_pre < 0.25 * n_pairs, "the fp32 pre-test should reject most (ray, triangle) pairs that miss"


def test_device_intersect_selftest_edge_cases(gpu):
    """ragged counts (not a multiple of the 64-case block), a single case, n = 0"""
    pr = np.load(GOLD + "/primitives.npz", allow_pickle=False)
    rays, cen, rad = pr["sph_ray"], pr["sph_center"], pr["sph_radius"]
    full = _selftest_intersect(0, rays, np.concatenate([cen, rad[:, None]], axis=1), 64.0)
    for n in (1, 63, 65, 130):
        part = _selftest_intersect(0, rays[:n], np.concatenate([cen[:n], rad[:n, None]], axis=1), 64.0)
        assert np.array_equal(part[0], full[0][:n]) and np.array_equal(part[1], full[1][:n])
    _selftest_intersect(0, rays[:0], np.zeros((0, 4)), 64.0)


def test_two_cameras_of_one_scene_on_two_streams(gpu):
    """The camera-dependent tables (filter thresholds, fp32 hierarchy boxes: functions of near_R) are
    kept per (scene, near_R) and immutable once built, so launches of ONE scene with different cameras
    on different streams cannot disturb each other: concurrent == sequential, bit for bit.  More
    cameras than table sets (8) exercises the recycling path."""
    import torch
    from rt_amd import scene as S
    for cfg, w, h, spp in [(4, 160, 96, 8), (5, 96, 56, 2)]:
        sc = S.build_scene(cfg, w, h, spp)
        gs = gpu.GpuScene(sc)
        total = gpu.n_tiles(w, h)
        cams = [S.make_camera(w, h, (3.0 * k - 12, 2.0 * (k % 3), 50.0 - 4 * k), (0, 0, 0)) for k in range(11)]
        seq = []
        for cam in cams:
            t, t8, st = gs.render_tiles(SEED, 0, 1, total, camera=cam)
            torch.cuda.synchronize()
            seq.append((t.clone(), t8.clone(), st.clone()))
        assert not torch.equal(seq[0][0], seq[1][0])
        gs.close()
        gs = gpu.GpuScene(sc)  # fresh table cache
        streams = [torch.cuda.Stream() for _ in range(3)]
        out = []
        for k, cam in enumerate(cams):
            with torch.cuda.stream(streams[k % 3]):
                out.append(gs.render_tiles(SEED, 0, 1, total, camera=cam))
        torch.cuda.synchronize()
        for (t, t8, st), (a, a8, ast) in zip(out, seq):
            assert torch.equal(t, a) and torch.equal(t8, a8) and torch.equal(st, ast)
        gs.close()
        # every camera on TWO streams at once (one table set read from two streams: it keeps a last-read event per
        # reader stream, and may only be recycled when both readers are done), again more cameras than table sets
        gs = gpu.GpuScene(sc)
        out = []
        for k, cam in enumerate(cams):
            pair = []
            for j in (0, 1):
                with torch.cuda.stream(streams[(k + j) % 3]):
                    pair.append(gs.render_tiles(SEED, 0, 1, total, camera=cam))
            out.append(pair)
        torch.cuda.synchronize()
        for pair, (a, a8, ast) in zip(out, seq):
            for t, t8, st in pair:
                assert torch.equal(t, a) and torch.equal(t8, a8) and torch.equal(st, ast)
        gs.close()
        sc.free()


def test_refraction_on_the_pooled_body(gpu, pt):
    """Small sphere scenes with M_REFRACTION render on the pooled body since round 4 (pt_render_tiles_refr_pool): pixel sums
    without a bound on the terms (windowed integer sums: order-free, so partitions and repeated runs are bit-identical),
    pending second children in stacks that travel with a path.  Against the oracle on scenes where the weights run away
    (camera inside a glass sphere: 'fresnel' 7.3 per hit, negative 'kt'), against the static kernel it replaces, and the
    launches it hands back to the static kernel (a sample x depth product the windowed sums could not hold)."""
    import os
    import subprocess
    import sys
    import torch
    from rt_amd import abi, scene as S
    from util import glass_scene
    sc = glass_scene(96, 64, 12, 7)
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name() == "pt_render_tiles_refr_pool"
    total = gpu.n_tiles(sc.width, sc.height)
    ref_t, ref_t8, ref_s = gs.render_tiles(SEED, 0, 1, total)
    again_t, again_t8, again_s = gs.render_tiles(SEED, 0, 1, total)
    torch.cuda.synchronize()
    assert torch.equal(again_t, ref_t) and torch.equal(again_t8, ref_t8) and torch.equal(again_s, ref_s)
    for r in range(3):   # any partition of the tiles: the same bits
        first, stride, count = gpu.rank_tiles(sc.width, sc.height, r, 3)
        t, t8, _ = gs.render_tiles(SEED, first, stride, count)
        torch.cuda.synchronize()
        assert torch.equal(t, ref_t[first::stride][:count]) and torch.equal(t8, ref_t8[first::stride][:count])
    gs.close()
    _full(gpu, pt, sc, hdr=True)
    # the camera inside a white glass sphere inside a lit room: every path starts with refractions from inside
    objs = [dict(flags=abi.M_REFRACTION, radius=3.0, center=(0, 0, 0), color=(1, 1, 1)),
            dict(flags=abi.M_REFRACTION | abi.M_CHECKERED, radius=1.0, center=(0.5, 0.2, -1.2), color=(0.9, 0.95, 1.0)),
            dict(flags=abi.M_DEFAULT, radius=1e4, center=(0, -10006.0, 0), color=(0.7, 0.7, 0.7)),
            dict(flags=abi.M_REFLECTION, radius=2.0, center=(5, -1, -4), color=(1, 1, 1)),
            dict(flags=abi.M_DEFAULT, radius=4.0, center=(-3, 9, 2), color=(1, 1, 1), emission=(6, 5, 4))]
    for depth in (5, 10):
        inside = S.custom_scene(objs, 64, 40, 6, depth, (0.3, 0.1, 0.8), (0.4, 0.3, -2.0))
        st = _full(gpu, pt, inside, hdr=True)
        assert st["rays"] > 64 * 40 * 6 * 4      # the trees really branch
    # the static kernel (the development build's RT_HIP_KERNEL_VARIANT=7, read once per process: a child) gives the same image to float rounding and the same counters
    code = ("import sys, json, torch; sys.path[:0] = [%r, %r]; from rt_amd import gpu as G; from util import glass_scene; "
            "sc = glass_scene(96, 64, 12, 7); gs = G.GpuScene(sc); img, img8, st = gs.render_image(%d); "
            "print(json.dumps({'k': gs.kernel_name(), 'st': st, 'sum': float(img.double().sum()), 'img8': int(img8.long().sum())}))"
            % (os.path.join(ROOT, "raytracer.c_amd"), os.path.join(ROOT, "tests"), SEED))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RT_HIP_KERNEL_VARIANT="7", RT_HIP_SHIM_PATH=DEV_LIB),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
    import json
    other = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    gs = gpu.GpuScene(sc)
    img, img8, st = gs.render_image(SEED)
    assert other["k"] == "pt_render_tiles_refr" and other["st"] == st
    assert abs(other["sum"] - float(img.double().sum())) <= 1e-6 * abs(other["sum"]) and abs(other["img8"] - int(img8.long().sum())) <= 64
    # a small mesh next to the glass (and a glass mesh): the pooled refraction kernel of the flat-filter scene class
    tri_objs = objs[1:]
    quad = [[(-4, -2, -6, 0, 0), (4, -2, -6, 1, 0), (0, 5, -6, 0, 1)], [(-4, -2, -6, 0, 0), (0, 5, -6, 0, 1), (-5, 4, -5, 1, 1)]]
    msc = S.custom_scene(tri_objs, 64, 40, 6, 6, (0.3, 1.0, 9.0), (0.4, 0.3, -2.0),
                         meshes=[dict(flags=abi.M_REFRACTION, color=(1, 1, 1), triangles=quad)])
    gsm = gpu.GpuScene(msc)
    assert gsm.kernel_name() == "pt_render_tiles_tri_refr_pool", gsm.kernel_name()
    gsm.close()
    _full(gpu, pt, msc, hdr=True)
    # a launch whose samples x 2^(max_depth + 2) exceeds 2^30 goes to the static kernel inside the same library: still the oracle's image
    # (the glass of this scene is out of reach, so that depth 29 does not mean 2^29 rays a sample on the CPU side)
    far_glass = [dict(flags=abi.M_REFRACTION, radius=1.0, center=(-3.0e6, 5.0e6, 0), color=(0.9, 0.9, 0.9))] + objs[2:]
    deep = S.custom_scene(far_glass, 32, 24, 2, 29, (0.3, 4.0, 12.0), (0.4, 0.3, -2.0))   # 2 x 2^31 > 2^30
    gsd = gpu.GpuScene(deep)
    assert gsd.kernel_name() == "pt_render_tiles_refr_pool"   # by scene; the launcher hands this launch to the static kernel
    gsd.close()
    _full(gpu, pt, deep, hdr=True)
    gs.close()


def test_nan_samples_poison_the_pixel_in_every_kernel_family(gpu):
    """A sample that turns NaN must make its pixel NaN (float) / 255 (byte: CLAMP(NaN) = 1,
    raytracer.c:218) whichever kernel family renders the scene: the pooled kernels sum samples as
    integers and flag NaN samples apart (finish_pixels), the static ones sum in fp64.  The scene: the
    camera sits at the centre of a sphere so small that the hit point rounds onto the centre, the
    normal is normalize(0) = 0 * inf = NaN.  (The reference itself stops there: assert(m > 0) in
    vec3_normalize, vector.h:56 -- which is why this test has no oracle side.)"""
    import torch
    from rt_amd import abi, scene as S
    c = (1.0e9, 1.0e9, 1.0e9)
    base = [dict(flags=abi.M_DEFAULT, radius=2.0e-8, center=c, color=(0.5, 0.4, 0.3), emission=(0.3, 0.2, 0.1)),
            dict(flags=abi.M_DEFAULT, radius=5.0, center=(0, 0, 0), color=(0.7, 0.7, 0.7))]
    glass = dict(flags=abi.M_REFRACTION, radius=1.0, center=(-3.0e9, 5.0e9, 0), color=(0.9, 0.9, 0.9))  # never reached
    results = []
    for objs, chunks in ((base, 1), (base, 2), (base + [glass], 1)):
        sc = S.custom_scene(objs, 24, 16, 2, 4, c, (0, 0, 0))
        gs = gpu.GpuScene(sc)
        assert ("_refr" in gs.kernel_name()) == (len(objs) == 3)
        total = gpu.n_tiles(24, 16)
        t, t8, st = gs.render_tiles(SEED, 0, 1, total, chunks=chunks, samples=2 if chunks == 1 else 4)
        torch.cuda.synchronize()
        results.append((t.cpu().numpy(), t8.cpu().numpy(), st.cpu().tolist()))
        gs.close()
    pooled, chunked, static = results
    nan = np.isnan(pooled[0])
    assert nan.any() and not nan.all(), "the scene should give both NaN and finite pixels"
    assert (pooled[1][nan] == 255).all()
    assert np.array_equal(nan, np.isnan(static[0])), "pooled and static kernels disagree on which pixels are NaN"
    assert np.allclose(pooled[0][~nan], static[0][~nan], rtol=1e-6, atol=0) and np.array_equal(pooled[1], static[1])
    assert pooled[2][:2] == static[2][:2]  # rays, casts (tests differ: the static scene has one more sphere)
    # chunked render of 4 spp: flags travel through the workspace; more samples, more NaN pixels
    nan4 = np.isnan(chunked[0])
    assert nan4.any() and (chunked[1][nan4] == 255).all() and (nan4 | ~nan).all()


def test_render_image_reuses_its_context_between_frames(gpu, pt):
    """rt_hip_render_image() (what render() of the C host calls per frame) keeps scenes, streams,
    buffers and communicators while device count, image size and scene bytes are unchanged"""
    from rt_amd import abi, scene as S
    shim = abi.load_shim()
    shim.rt_hip_release_cache()
    b0 = shim.rt_hip_cache_builds()
    sc = S.build_scene(2, 72, 40, 4)
    a = gpu.render_image_host(sc, SEED)
    b = gpu.render_image_host(sc, SEED)
    assert shim.rt_hip_cache_builds() == b0 + 1, "second frame of the same scene must not rebuild"
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    c = gpu.render_image_host(sc, SEED + 1)   # another seed: same context
    assert shim.rt_hip_cache_builds() == b0 + 1 and not np.array_equal(a[0], c[0])
    sc2 = S.build_scene(2, 72, 40, 4)
    sc2.objects[1].radius *= 1.5              # one byte of the scene changes: rebuilt, and rendered correctly
    d = gpu.render_image_host(sc2, SEED)
    assert shim.rt_hip_cache_builds() == b0 + 2
    mean, rgb8, ost = pt.render_pixels(sc2, SEED)
    assert_parity(d[0], d[1], d[2], mean, rgb8, ost, what="rebuilt context")
    e = gpu.render_image_host(sc, SEED, samples=8)   # other spp, same scene as the first: rebuilt once more (scene differs from sc2)
    assert shim.rt_hip_cache_builds() == b0 + 3
    shim.rt_hip_release_cache()
    f = gpu.render_image_host(sc, SEED)
    assert np.array_equal(a[0], f[0]) and shim.rt_hip_cache_builds() == b0 + 4


def test_hierarchy_kernel_without_its_workspace(gpu, pt):
    """a hierarchy scene whose ring workspace cannot be allocated (rt_hip_selftest_fail_alloc: the allocation behaves as
    failed) takes the pick table's park = NO row -- the lane-waiting kernel, every ray that can reach the mesh walked from its
    lane's registers -- says so, and renders the same bits"""
    import torch
    from rt_amd import abi, scene as S
    shim = abi.load_shim()
    sc = S.build_scene(5, 72, 40, 6)
    shim.rt_hip_selftest_fail_alloc(abi.FAIL_ALLOC_PARK_WS)
    try:
        gs = gpu.GpuScene(sc)
        assert gs.kernel_name() == "pt_render_tiles_tri_big"
        img, img8, st = gs.render_image(SEED)
        assert gs.last_launch_kernel() == "pt_render_tiles_tri_big"
    finally:
        shim.rt_hip_selftest_fail_alloc(0)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what="config 5 without ring workspace")
    assert gs.kernel_name() == "pt_render_tiles_tri_big", "a scene that met no workspace keeps saying so"
    gs.close()
    gs = gpu.GpuScene(sc)  # and with it: the same image, bit for bit
    assert gs.kernel_name() == "pt_render_tiles_tri_queued_sph"  # config 5's mesh is round: the probe is its bounding sphere
    img2, img82, st2 = gs.render_image(SEED)
    assert gs.last_launch_kernel() == "pt_render_tiles_tri_queued_sph"
    assert torch.equal(img, img2) and torch.equal(img8, img82) and st == st2
    gs.close()
    # the checkered form of the fallback
    sc.objects[0].flags |= abi.M_CHECKERED
    shim.rt_hip_selftest_fail_alloc(abi.FAIL_ALLOC_PARK_WS)
    try:
        gs = gpu.GpuScene(sc)
        img, img8, st = gs.render_image(SEED)
        assert gs.last_launch_kernel() == "pt_render_tiles_tri_big_chk"
    finally:
        shim.rt_hip_selftest_fail_alloc(0)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what="config 5, checkered wall, without ring workspace")
    gs.close()
    sc.free()


def test_glass_mesh_when_the_wide_pending_ray_pool_cannot_be_had(gpu, pt):
    """the parked-walk refraction kernels want 4 x 512 pending-ray stacks per pool slot; where that allocation fails the
    ordinary pool and the static kernel of the family render the scene (the table's fit = NO row) -- and the failed hipMalloc
    must not surface as the launch's error (round-4 advisor finding: its sticky error used to)"""
    from rt_amd import abi, scene as S
    shim = abi.load_shim()
    shim.rt_hip_release_cache()          # drops the pending-ray pool: the next launch has to allocate one
    sc = S.build_scene(5, 72, 40, 6, max_depth=5)
    sc.meshes[0].flags = abi.M_REFRACTION
    shim.rt_hip_selftest_fail_alloc(abi.FAIL_ALLOC_WIDE_PEND)
    try:
        gs = gpu.GpuScene(sc)
        assert gs.kernel_name() == "pt_render_tiles_tri_big_refr"
        img, img8, st = gs.render_image(SEED)
        assert gs.last_launch_kernel() == "pt_render_tiles_tri_big_refr"
    finally:
        shim.rt_hip_selftest_fail_alloc(0)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what="glass mesh, narrow pool", hdr=True)
    img2, img82, st2 = gs.render_image(SEED)   # the pool can be had now: the parked-walk kernel, same counters, same image to rounding
    assert gs.last_launch_kernel() == "pt_render_tiles_tri_queued_refr_sph"
    assert st2 == st
    assert_parity(img2.cpu().numpy(), img82.cpu().numpy(), st2, mean, rgb8, ost, what="glass mesh, wide pool", hdr=True)
    gs.close()
    sc.free()


@pytest.mark.parametrize("what", ["wall", "mesh"])
def test_checkered_materials_with_a_round_mesh_take_the_sphere_probe_kernel(gpu, pt, what):
    """config 5's scene with M_CHECKERED on a wall sphere, or on the mesh itself (then hit.u / hit.v follow every passing
    triangle: no pruning by the closest hit, no hull-facet rule): the parked-walk kernel whose probe is the bounding sphere
    alone (the mesh is round), with the checker code -- pt_render_tiles_tri_queued_chk_sph; frame and counters = oracle"""
    from rt_amd import abi, scene as S
    sc = S.build_scene(5, 80, 48, 4)
    if what == "wall":
        sc.objects[0].flags |= abi.M_CHECKERED
    else:
        sc.meshes[0].flags |= abi.M_CHECKERED
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name() == "pt_render_tiles_tri_queued_chk_sph", gs.kernel_name()
    img, img8, st = gs.render_image(SEED)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what=f"config 5, checkered {what}")
    gs.close()
    sc.free()


@pytest.mark.parametrize("case", ["config5", "convex", "nested", "deep", "deeper"])
def test_glass_mesh_through_the_hierarchy_on_the_parked_walk_body(gpu, pt, case):
    """hierarchy scenes with M_REFRACTION (pt_render_tiles_tri_queued_refr[_sph], end of round 4): windowed pixel sums, pending
    second children whose stack id travels with the path through the waiting list AND the ring, the hull-facet rule for both
    children of a refractive hit -- frames and counters equal the oracle's linear scan and its recursion (raytracer.c:514-529).
    config5: the 10,240-triangle sphere turned to glass (+ a glass sphere); convex: a glass polyhedron with a second body
    inside it (children that leave a hull facet of the outer body must still find the inner one where the rule does not
    apply); nested: depth 12 and many samples per pixel (long pools: ids are taken and given back thousands of times); deep:
    max_depth 29 -- the windowed sums hold ONE sample per chunk there (samples_per_chunk x 2^(max_depth + 1) <= 2^30), so the
    frame is rendered in as many sample chunks as it has samples; deeper: max_depth 30 fits no chunking, the launch takes
    the static kernel of the family (which finds the wide pool's slots laid out for it as well)"""
    from rt_amd import abi, scene as S
    from util import convex_body_scene, fixed_point_floor
    if case == "config5":
        sc = S.build_scene(5, 96, 54, 6, 8)
        sc.meshes[0].flags = abi.M_REFRACTION
        sc.objects[sc.n_objects - 1].flags = abi.M_REFRACTION
    elif case == "convex":
        sc = convex_body_scene(1, 72, 44, 12)[0]       # (odd seed: a second body inside the first)
        sc.max_depth = 6
        sc.meshes[0].flags = abi.M_REFRACTION
    elif case == "nested":
        sc = convex_body_scene(3, 40, 24, 96)[0]
        sc.max_depth = 12
        for m in range(sc.n_meshes):
            sc.meshes[m].flags = abi.M_REFRACTION | (abi.M_CHECKERED if m else 0)
    else:
        sc = convex_body_scene(5, 32, 20, 2)[0]
        sc.max_depth = 29 if case == "deep" else 30
        sc.meshes[0].flags = abi.M_REFRACTION
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name().startswith("pt_render_tiles_tri_queued_refr"), gs.kernel_name()
    img, img8, st = gs.render_image(SEED)
    # what the launch itself took (rt_hip_last_launch_kernel): the scene's row, unless the launch's own facts name another
    assert gs.last_launch_kernel() == ("pt_render_tiles_tri_big_refr" if case == "deeper" else gs.kernel_name())
    if case == "deep":
        assert gs.suggest_chunks(gpu.n_tiles(sc.width, sc.height)) == sc.samples == 2
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what=f"glass mesh, {case}", hdr=True,
                  abs_floor=fixed_point_floor(sc))
    # any tile partition gives the same bits (windowed integer sums: order-free)
    total = gpu.n_tiles(sc.width, sc.height)
    import torch
    t_all, t8_all, _ = gs.render_tiles(SEED, 0, 1, total)
    t_odd, t8_odd, _ = gs.render_tiles(SEED, 1, 2, total // 2)
    torch.cuda.synchronize()
    assert torch.equal(t_all[1::2][: total // 2], t_odd[: total // 2]) and torch.equal(t8_all[1::2][: total // 2], t8_odd[: total // 2])
    gs.close()


@pytest.mark.parametrize("seed", [3, 4])
def test_hierarchy_builders_give_the_same_frame(gpu, pt, seed, tmp_path):
    """the hierarchy only decides WHICH triangles get the exact test: the surface-area builder (round 4; 64 bins, within the
    least depth leaves of 15 allow) and the median builder of rounds 1-3 (the development build's RT_HIP_BVH_MEDIAN=1: a child
    process on librt_hip_dev.so) must render the same bits -- on a lopsided mesh, where they differ most: three clusters of
    triangles whose sizes range over two decades, one cluster holding most of them (so the depth cap rules out most area
    splits near the root), plus duplicated and degenerate triangles; and the frame equals the oracle's linear scan"""
    import os
    import subprocess
    import sys
    from util import lopsided_mesh_scene
    sc = lopsided_mesh_scene(seed)
    assert sc.n_triangles > 256
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name().startswith("pt_render_tiles_tri_queued")
    img, img8, st = gs.render_image(SEED)
    gs.close()
    out = str(tmp_path / "median.npz")
    code = ("import sys, numpy as np, torch; sys.path[:0] = [%r, %r]; from rt_amd import gpu as G; from util import lopsided_mesh_scene; "
            "sc = lopsided_mesh_scene(%d); gs = G.GpuScene(sc); img, img8, st = gs.render_image(%d); "
            "np.savez(%r, img=img.cpu().numpy(), img8=img8.cpu().numpy(), st=np.array([st['rays'], st['casts'], st['tests'], st['samples']]))"
            % (os.path.join(ROOT, "raytracer.c_amd"), os.path.join(ROOT, "tests"), seed, SEED, out))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RT_HIP_BVH_MEDIAN="1", RT_HIP_SHIM_PATH=DEV_LIB),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
    m = np.load(out)
    assert np.array_equal(m["img"], img.cpu().numpy()) and np.array_equal(m["img8"], img8.cpu().numpy())
    assert m["st"].tolist() == [st["rays"], st["casts"], st["tests"], st["samples"]]
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what="lopsided mesh, surface-area hierarchy")


def test_small_mesh_kernel_partitions_and_chunks_are_bit_invariant(gpu, pt):
    """pt_render_tiles_tri (fp32 triangle pre-test, two candidate loops, throughput parked in LDS across the
    scan, radiance flushed per trip): interleaved tile subsets and sample chunks reproduce the full render
    bit for bit, and the full render matches the oracle"""
    import torch
    from rt_amd import scene as S
    sc = S.build_scene(3, 136, 80, 12)
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name() == "pt_render_tiles_tri"
    total = gpu.n_tiles(sc.width, sc.height)
    ref_t, ref_t8, ref_s = gs.render_tiles(SEED, 0, 1, total)
    torch.cuda.synchronize()
    for chunks in (2, 3, 12):
        t, t8, s = gs.render_tiles(SEED, 0, 1, total, chunks=chunks)
        torch.cuda.synchronize()
        assert torch.equal(t, ref_t) and torch.equal(t8, ref_t8) and torch.equal(s, ref_s), chunks
    full, full8 = gs.untile(ref_t, ref_t8, 0, 1, total)
    image, image8 = torch.zeros_like(full), torch.zeros_like(full8)
    tot = torch.zeros(4, dtype=torch.int64, device=full.device)
    for r in range(3):
        first, stride, count = gpu.rank_tiles(sc.width, sc.height, r, 3)
        t, t8, s = gs.render_tiles(SEED, first, stride, count)
        gs.untile(t, t8, first, stride, count, image, image8)
        tot += s
    torch.cuda.synchronize()
    assert torch.equal(image, full) and torch.equal(image8, full8) and torch.equal(tot, ref_s)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    st = ref_s.cpu().tolist()
    assert_parity(full.cpu().numpy(), full8.cpu().numpy(), dict(rays=st[0], casts=st[1], tests=st[2]), mean, rgb8, ost,
                  what="config 3 reduced")
    gs.close()
    sc.free()


@pytest.mark.parametrize("n_tris", [6, 300])
def test_mesh_only_scene_without_spheres(gpu, pt, n_tris):
    """no spheres at all: the flat filter (small mesh) resp. the parked-walk kernel's sphere scan (large
    mesh) has nothing to do, every hit comes from triangles; an emissive mesh lights a diffuse one"""
    from rt_amd import abi, scene as S
    rng = np.random.default_rng(77 + n_tris)
    tris = []
    for _ in range(n_tris):
        c = rng.uniform(-5, 5, 3)
        tris.append([tuple(c + rng.normal(0, 1.5, 3)) + (0.0, 0.0) for _ in range(3)])
    floor = [[(-30, -6, -30, 0, 0), (30, -6, -30, 1, 0), (0, -6, 40, 0, 1)]]
    meshes = [dict(flags=abi.M_DEFAULT, color=(0.7, 0.6, 0.5), triangles=tris),
              dict(flags=abi.M_DEFAULT, color=(1, 1, 1), emission=(3, 3, 3), triangles=floor)]
    sc = S.custom_scene([], 48, 32, 4, 4, (0, 2, 25), (0, 0, 0), meshes=meshes)
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name() == ("pt_render_tiles_tri" if n_tris < 250 else "pt_render_tiles_tri_queued")  # a soup is not round
    gs.close()
    st = _full(gpu, pt, sc)
    assert st["tests"] == st["casts"] * (n_tris + 1) and st["casts"] > 48 * 32 * 4
    _full(gpu, pt, sc, integrator="whitted")


def test_xcc_id_partitions_the_workgroups(gpu):
    """the parked-walk kernels key their workspace pools by HW_REG_XCC_ID of the running wave: on an
    MI355X (8 XCDs) a 2,048-workgroup launch must report every id in 0..7 (and none beyond), each with a fair share"""
    import ctypes as C
    from rt_amd import abi
    shim = abi.load_shim()
    counts = (C.c_uint32 * 16)()
    assert shim.rt_hip_selftest_xcc(2048, counts, 0) == 0, shim.rt_hip_last_error()
    got = list(counts)
    assert sum(got) == 2048 and sum(got[8:]) == 0, got
    name = C.create_string_buffer(128)
    cus = C.c_int(0)
    shim.rt_hip_device_info(0, name, 128, C.byref(cus))
    if cus.value == 256:  # the full chip (SPX mode): 8 XCDs
        assert all(c >= 2048 // 16 for c in got[:8]), got
    else:
        assert max(got) > 0


@pytest.mark.parametrize("cfg,spp", [(4, 0), (3, 0), (5, 256)])
def test_full_size_frames_are_partition_and_chunk_invariant(gpu, cfg, spp):
    """BASELINE.json's full sizes through size-independent properties (the oracle cannot render these
    frames): config 4 (1920x1080 x 1024 spp) and config 3 (x 256 spp) as they are, config 5 at its full
    3840x2160 with 256 of its 4096 spp.  One launch == the 8-rank interleaved partition with each rank's
    suggested sample chunks, bit for bit (float image, bytes, every counter); counters are self-consistent."""
    import torch
    from rt_amd import scene as S
    sc = S.build_scene(cfg, samples=spp or None)
    gs = gpu.GpuScene(sc)
    total = gpu.n_tiles(sc.width, sc.height)
    t, t8, st = gs.render_tiles(SEED, 0, 1, total)
    full, full8 = gs.untile(t, t8, 0, 1, total)
    torch.cuda.synchronize()
    s = st.cpu().tolist()
    n_prims = sc.n_objects + sc.n_triangles
    assert s[3] == sc.width * sc.height * sc.samples and s[2] == s[1] * n_prims
    assert s[3] <= s[1] <= s[0] <= s[3] * (sc.max_depth + 2)
    image, image8 = torch.zeros_like(full), torch.zeros_like(full8)
    tot = torch.zeros(4, dtype=torch.int64, device=full.device)
    world = 8
    for r in range(world):
        first, stride, count = gpu.rank_tiles(sc.width, sc.height, r, world)
        chunks = gs.suggest_chunks(count)
        tr, tr8, sr = gs.render_tiles(SEED, first, stride, count, chunks=chunks)
        gs.untile(tr, tr8, first, stride, count, image, image8)
        tot += sr
    torch.cuda.synchronize()
    assert torch.equal(image, full) and torch.equal(image8, full8)
    assert torch.equal(tot, st)
    assert torch.isfinite(full).all() and float(full.max()) > 0.5
    gs.close()
    sc.free()


def _icosphere(radius, center, levels):
    """triangles (each 3 x (x, y, z)) of a subdivided icosahedron: 20 * 4**levels faces, vertices on the sphere"""
    import math
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]

    def unit(p):
        n = math.sqrt(sum(x * x for x in p))
        return tuple(x / n for x in p)
    tris = [tuple(unit(v[i]) for i in face) for face in f]
    for _ in range(levels):
        nxt = []
        for a, b, c in tris:
            ab, bc, ca = (unit(tuple((x + y) / 2 for x, y in zip(p, q))) for p, q in ((a, b), (b, c), (c, a)))
            nxt += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        tris = nxt
    return [[tuple(center[k] + radius * p[k] for k in range(3)) for p in tri] for tri in tris]


@pytest.mark.parametrize("cam", [(0, 4, 30), (0, 0.5, 4.2), (0.3, 0.2, 0.1), (0, 0, 3.9999), (0, 9, 0.5), (55, 30, 40)])
def test_round_mesh_probe_from_outside_near_and_inside(gpu, pt, cam):
    """the probe that decides which rays are parked for a hierarchy walk tests the triangles' bounding sphere
    (bvh_probe, conservative fp32): a round mesh (320-face icosphere of radius 4, where that sphere is tight) seen
    from far away, from just outside its surface, from its centre region, from a point on the bounding sphere
    itself to rounding, from above, and from beyond the scene -- frames and counters equal the oracle's, which
    scans every triangle"""
    from rt_amd import abi, scene as S
    meshes = [dict(flags=abi.M_DEFAULT, color=(0.8, 0.5, 0.3), triangles=_icosphere(4.0, (0, 0, 0), 2))]
    objs = [dict(flags=abi.M_DEFAULT, radius=1e4, center=(0, -10006.0, 0), color=(0.7, 0.7, 0.7)),
            dict(flags=abi.M_REFLECTION, radius=3.0, center=(8, -1, 2), color=(1, 1, 1)),
            dict(flags=abi.M_DEFAULT, radius=2.0, center=(-7, 1, 5), color=(0.3, 0.4, 0.9)),
            dict(flags=abi.M_DEFAULT, radius=6.0, center=(0, 22, 0), color=(1, 1, 1), emission=(4, 4, 4))]
    sc = S.custom_scene(objs, 64, 40, 4, 6, cam, (0, 0, 0) if cam != (0.3, 0.2, 0.1) else (5, 1, 5), meshes=meshes)
    assert sc.n_triangles == 320
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name() == "pt_render_tiles_tri_queued_sph"
    gs.close()
    st = _full(gpu, pt, sc)
    assert st["tests"] == st["casts"] * (4 + 320)
    _full(gpu, pt, sc, integrator="whitted")


def _hull_scene(tris, cam=(0, 5, 30), flags=None, extra=(), samples=4, size=(64, 40)):
    from rt_amd import abi, scene as S
    meshes = [dict(flags=abi.M_DEFAULT if flags is None else flags, color=(0.8, 0.6, 0.4), triangles=tris)] + list(extra)
    objs = [dict(flags=abi.M_DEFAULT, radius=1e4, center=(0, -10006.0, 0), color=(0.7, 0.7, 0.7)),
            dict(flags=abi.M_REFLECTION, radius=3.0, center=(9, -1, 2), color=(1, 1, 1)),
            dict(flags=abi.M_DEFAULT, radius=6.0, center=(0, 22, 0), color=(1, 1, 1), emission=(4, 4, 4))]
    return S.custom_scene(objs, size[0], size[1], samples, 8, cam, (0, 0, 0), meshes=meshes)


def test_hull_facets_are_marked_and_bounces_off_them_skip_the_walk(gpu, pt):
    """HULL FACETS (pt_build_hull_flags): every triangle of the scene on the inner side of the facet's plane.  A
    bounce that leaves one on its outer side is not walked through the hierarchy; images and counters must still
    equal the oracle's, which tests every triangle for every ray.  Cases: a convex mesh with outward normals (all
    facets marked '+'), the same wound the other way ('-': diffuse bounces go inward and must be walked), a mirror
    convex mesh, two convex meshes side by side (only the facets whose plane clears the other mesh stay marked), a
    convex mesh with needle-shaped facets (not marked: the error bound behind the margin needs a decent shape)."""
    from rt_amd import abi
    ico = _icosphere(4.0, (0, 0, 0), 2)
    flipped = [[a, c, b] for a, b, c in ico]
    cases = []
    sc = _hull_scene(ico)
    cases.append((sc, "convex", lambda p, m: p + m == 320 and (p == 320 or m == 320)))
    sc2 = _hull_scene(flipped)
    cases.append((sc2, "convex, wound the other way", lambda p, m: p + m == 320 and (p == 320 or m == 320)))
    cases.append((_hull_scene(ico, flags=abi.M_REFLECTION), "mirror", lambda p, m: p + m == 320))
    other = dict(flags=abi.M_DEFAULT, color=(0.3, 0.5, 0.9), triangles=_icosphere(3.0, (8.5, 0, 1), 2))
    cases.append((_hull_scene(ico, extra=[other]), "two meshes", lambda p, m: 0 < p + m < 640))
    # a long thin pyramid: four needle facets (apex angle ~0.3 degrees) and a square base
    apex, h = (0.0, 2.0, 0.0), 0.02
    base = [(-h, -2.0, -h), (h, -2.0, -h), (h, -2.0, h), (-h, -2.0, h)]
    needle = [[apex, base[(k + 1) % 4], base[k]] for k in range(4)] + [[base[0], base[1], base[2]], [base[0], base[2], base[3]]]
    big = dict(flags=abi.M_DEFAULT, color=(0.3, 0.5, 0.9), triangles=ico)  # enough triangles for the hierarchy kernels
    cases.append((_hull_scene([[tuple(8 + x if i == 0 else x for i, x in enumerate(v)) for v in t] for t in needle], extra=[big]),
                  "needles", None))
    signs = []
    for sc, what, check in cases:
        gs = gpu.GpuScene(sc)
        assert gs.kernel_name().startswith("pt_render_tiles_tri_queued"), what
        p, m = gs.hull_facets()
        if check is not None:
            assert check(p, m), (what, p, m)
        signs.append((p, m))
        gs.close()
        _full(gpu, pt, sc)
        _full(gpu, pt, sc, integrator="whitted")
    assert signs[0] == signs[1][::-1], "flipping the winding flips the side the stored normal is on"
    assert signs[0][0] + signs[0][1] == 320


def test_hull_facets_config5_and_walk_counts(gpu):
    """BASELINE configs[4]'s mesh is convex: all of its facets but the needles at the poles are hull facets"""
    from rt_amd import scene as S
    sc = S.build_scene(5, 64, 36, 2)
    gs = gpu.GpuScene(sc)
    p, m = gs.hull_facets()
    assert (p == 0 or m == 0) and 9000 <= p + m <= 10240, (p, m)
    gs.close()
    sc.free()


@pytest.mark.parametrize("seed", range(8))
def test_hull_fuzz_random_convex_meshes(gpu, pt, seed):
    """random convex polyhedra (scipy's hull of random points on an ellipsoid: ~400 facets of every shape), each
    facet wound at random (stored normals point in or out per facet), placed near or hundreds of units from the
    origin (the margin grows with the extent), diffuse or mirror, a second body inside or beside the first on odd
    seeds (its facets and the outer ones that do not clear it lose their marks): frames and counters equal the oracle's"""
    from util import convex_body_scene
    sc, n_outer = convex_body_scene(seed)
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name().startswith("pt_render_tiles_tri_queued")
    p, m = gs.hull_facets()
    if seed % 2 == 0:
        assert p + m >= 0.8 * n_outer and p > 0 and m > 0, (p, m, n_outer)  # needles aside, every facet; both windings
    else:
        assert p + m < sc.n_triangles
    gs.close()
    _full(gpu, pt, sc)


@pytest.mark.parametrize("seed", range(8))
def test_leading_walls_pruned_among_themselves(gpu, pt, seed):
    """rooms of 2-8 leading wall-sized spheres, with near-coincident and duplicated walls, the camera almost on or inside
    a wall: the kernels prune such walls among themselves by fp32 distance bounds before the exact tests (BigPrune) --
    images and counters must stay the oracle's; seeds 4-7 add a 300-triangle mesh, i.e. the parked-walk kernels"""
    from util import walls_scene
    sc = walls_scene(seed, with_mesh=seed >= 4)
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name().startswith("pt_render_tiles_tri_queued") if seed >= 4 else gs.kernel_name() == "pt_render_tiles"
    gs.close()
    _full(gpu, pt, sc, hdr=True)


def test_inside_a_mesh_every_ray_is_parked(gpu, pt):
    """the camera inside a closed, bumpy mesh shell (an icosphere of 1,280 faces with every other vertex pushed in: not
    convex, so no bounce is spared its walk), lit by a small sphere inside, 48 spp: every ray of every bounce is parked for
    a hierarchy walk, a wave's pool is thousands of paths -- the ring, the waiting list and the walk-as-swap at their
    busiest (a ring that could fill up would starve such a wave: PT_PARK_Q in pt_body_queued.h); frame and counters = oracle"""
    import math
    from rt_amd import abi, scene as S
    tris = []
    for tri in _icosphere(6.0, (0, 0, 0), 3):
        out = []
        for (x, y, z) in tri:
            r = 1.0 - 0.08 * (math.sin(5 * x) * math.sin(7 * y + 1.0) * math.sin(3 * z + 2.0) > 0)  # a deterministic dimple pattern
            out.append((x * r, y * r, z * r))
        tris.append(out)  # counter-clockwise seen from outside: the reference's normal formula, cross(v2 - v0, v1 - v0), points INTO the shell
    meshes = [dict(flags=abi.M_DEFAULT, color=(0.85, 0.8, 0.7), triangles=tris)]
    objs = [dict(flags=abi.M_DEFAULT, radius=0.8, center=(0.5, 2.0, 0.3), color=(1, 1, 1), emission=(9, 8, 7)),
            dict(flags=abi.M_REFLECTION, radius=1.0, center=(-2.0, -1.5, 1.0), color=(1, 1, 1))]
    sc = S.custom_scene(objs, 48, 32, 48, 7, (0.2, -0.3, -3.5), (0.5, 0.5, 2.0), meshes=meshes)
    assert sc.n_triangles == 1280
    gs = gpu.GpuScene(sc)
    assert gs.kernel_name().startswith("pt_render_tiles_tri_queued")
    gs.close()
    st = _full(gpu, pt, sc)
    assert st["casts"] > 3 * 48 * 32 * 48, "paths should bounce around inside the shell"


# ---- one small scene per row of the kernel pick table: every shipped kernel under an oracle comparison ---------------------
# (class of util.class_scene, integrator, allocation faults injected, the kernel the launch must take).  Together with
# tests/test_pick_table.py (the same rows without a GPU) and tests/test_zz_kernel_coverage.py (what this run launched).
_PW, _WP = 1, 2   # abi.FAIL_ALLOC_PARK_WS, FAIL_ALLOC_WIDE_PEND
PICK_ROWS = [
    (dict(n_packed=4), "path", 0, "pt_render_tiles"),
    (dict(n_packed=4, chk=True), "path", 0, "pt_render_tiles_chk"),
    (dict(n_packed=4, refr=True), "path", 0, "pt_render_tiles_refr_pool"),
    (dict(n_packed=4, refr=True, depth=30), "path", 0, "pt_render_tiles_refr"),
    (dict(n_packed=4, wide=True), "path", 0, "pt_render_tiles_big"),
    (dict(n_packed=4, wide=True, chk=True), "path", 0, "pt_render_tiles_big_chk"),
    (dict(n_packed=4, wide=True, refr=True), "path", 0, "pt_render_tiles_big_refr"),
    (dict(n_packed=4, tris=40), "path", 0, "pt_render_tiles_tri"),
    (dict(n_packed=4, tris=40, mesh_chk=True), "path", 0, "pt_render_tiles_tri_chk"),
    (dict(n_packed=4, tris=40, mesh_refr=True), "path", 0, "pt_render_tiles_tri_refr_pool"),
    (dict(n_packed=4, tris=40, mesh_refr=True, depth=30), "path", 0, "pt_render_tiles_tri_refr"),
    (dict(n_packed=4, tris=400), "path", 0, "pt_render_tiles_tri_queued"),
    (dict(n_packed=4, tris=400, chk=True), "path", 0, "pt_render_tiles_tri_queued_chk"),
    (dict(n_packed=4, tris=400, round_mesh=True), "path", 0, "pt_render_tiles_tri_queued_sph"),
    (dict(n_packed=4, tris=400, round_mesh=True, mesh_chk=True), "path", 0, "pt_render_tiles_tri_queued_chk_sph"),
    (dict(n_packed=4, tris=400, mesh_refr=True), "path", 0, "pt_render_tiles_tri_queued_refr"),
    (dict(n_packed=4, tris=400, round_mesh=True, mesh_refr=True), "path", 0, "pt_render_tiles_tri_queued_refr_sph"),
    (dict(n_packed=4, tris=400), "path", _PW, "pt_render_tiles_tri_big"),
    (dict(n_packed=4, tris=400, wide=True), "path", 0, "pt_render_tiles_tri_big"),
    (dict(n_packed=4, tris=400, chk=True), "path", _PW, "pt_render_tiles_tri_big_chk"),
    (dict(n_packed=4, tris=400, mesh_refr=True), "path", _WP, "pt_render_tiles_tri_big_refr"),
    # (the same kernel through sums that do not fit -- max_depth 30 -- is test_glass_mesh_through_the_hierarchy...[deeper]: a small
    # convex body; on this room's 400-triangle sheet a depth-30 tree takes two minutes)
    (dict(n_packed=120), "path", 0, "pt_render_tiles_pool_mem_s"),
    (dict(n_packed=120, chk=True), "path", 0, "pt_render_tiles_pool_mem_s_chk"),
    (dict(n_packed=120, refr=True), "path", 0, "pt_render_tiles_refr_pool_mem"),
    (dict(n_packed=300), "path", 0, "pt_render_tiles_pool_mem_s"),
    (dict(n_packed=300, refr=True), "path", 0, "pt_render_tiles_refr_pool_mem"),
    (dict(n_packed=300, refr=True, depth=30), "path", 0, "pt_render_tiles_mem"),
    (dict(n_packed=300, wide=True), "path", 0, "pt_render_tiles_pool_mem"),
    (dict(n_packed=300, wide=True, chk=True), "path", 0, "pt_render_tiles_pool_mem_chk"),
    (dict(n_packed=300, tris=60), "path", 0, "pt_render_tiles_pool_mem_tri"),
    (dict(n_packed=300, tris=60, chk=True), "path", 0, "pt_render_tiles_pool_mem_tri_chk"),
    (dict(n_packed=300, tris=400), "path", 0, "pt_render_tiles_tri_queued_mem"),
    (dict(n_packed=300, tris=400, mesh_chk=True), "path", 0, "pt_render_tiles_tri_queued_mem_chk"),
    (dict(n_packed=300, tris=400), "path", _PW, "pt_render_tiles_pool_mem_tri"),
    (dict(n_packed=300, tris=400, mesh_refr=True), "path", 0, "pt_render_tiles_mem"),
    (dict(n_packed=4, chk=True, refr=True), "whitted", 0, "pt_whitted_tiles"),
    (dict(n_packed=4, wide=True), "whitted", 0, "pt_whitted_tiles_big"),
    (dict(n_packed=4, tris=40), "whitted", 0, "pt_whitted_tiles_tri"),
    (dict(n_packed=4, tris=400), "whitted", 0, "pt_whitted_tiles_tri_big"),
    (dict(n_packed=4, glass2=True), "whitted", 0, "pt_whitted_tiles_mem"),
    (dict(n_packed=300, tris=60), "whitted", 0, "pt_whitted_tiles_mem"),
]


@pytest.mark.parametrize("cls,integrator,faults,kernel", PICK_ROWS,
                         ids=[f"{k}:{i}:{'+'.join(f'{a}={b}' for a, b in c.items())}{':fault%d' % f if f else ''}" for c, i, f, k in PICK_ROWS])
def test_every_row_of_the_pick_table_renders_the_oracle_s_frame(gpu, pt, cls, integrator, faults, kernel):
    from rt_amd import abi
    from util import class_scene, fixed_point_floor
    shim = abi.load_shim()
    sc = class_scene(**cls)
    if faults & _WP:
        shim.rt_hip_release_cache()      # no pending-ray pool yet: the launch has to ask for the wide one
    shim.rt_hip_selftest_fail_alloc(faults)
    try:
        gs = gpu.GpuScene(sc)
        assert gs.kernel_name(integrator) == kernel or cls.get("depth") == 30   # (the launch's own facts: below)
        img, img8, st = gs.render_image(SEED, integrator=integrator)
        assert gs.last_launch_kernel() == kernel
    finally:
        shim.rt_hip_selftest_fail_alloc(0)
    mean, rgb8, ost = pt.render_pixels(sc, SEED, integrator=integrator)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what=f"{kernel} {cls}", hdr=True,
                  abs_floor=fixed_point_floor(sc))
    gs.close()
    sc.free()


@pytest.mark.parametrize("kind", ["spheres", "streamed", "small_mesh", "glass_mesh"])
def test_refraction_kernels_take_sample_chunks(gpu, pt, kind):
    """round 5: the M_REFRACTION forms of the pooled and parked-walk kernels split a tile's samples over workgroups like the
    others -- each chunk's windowed sums (win_add) are carry-normalised and merged as integers, pt_resolve_tiles normalises the
    total: the frame is the one-chunk frame BIT FOR BIT for every chunk count and tile partition, and equals the oracle's"""
    import torch
    from rt_amd import abi, scene as S
    from util import class_scene, fixed_point_floor, glass_scene
    if kind == "spheres":
        sc, kernel = glass_scene(72, 48, 22, 6), "pt_render_tiles_refr_pool"          # 22 spp: chunks of unequal size
    elif kind == "streamed":
        sc, kernel = class_scene(n_packed=120, refr=True, width=56, height=40, samples=10), "pt_render_tiles_refr_pool_mem"
    elif kind == "small_mesh":
        sc, kernel = class_scene(n_packed=4, tris=40, mesh_refr=True, samples=9), "pt_render_tiles_tri_refr_pool"
    else:
        sc = S.build_scene(5, 96, 54, 12, 6)
        sc.meshes[0].flags = abi.M_REFRACTION
        kernel = "pt_render_tiles_tri_queued_refr_sph"
    gs = gpu.GpuScene(sc)
    total = gpu.n_tiles(sc.width, sc.height)
    ref_t, ref_t8, ref_s = gs.render_tiles(SEED, 0, 1, total)
    torch.cuda.synchronize()
    assert gs.last_launch_kernel() == kernel
    for chunks in (2, 3, 7, sc.samples):
        t, t8, s = gs.render_tiles(SEED, 0, 1, total, chunks=chunks)
        torch.cuda.synchronize()
        assert gs.last_launch_kernel() == kernel
        assert torch.equal(t, ref_t) and torch.equal(t8, ref_t8), (kind, chunks)
        assert torch.equal(s, ref_s), (kind, chunks, s.tolist(), ref_s.tolist())
    # a strided partition in three chunks
    t_odd, t8_odd, _ = gs.render_tiles(SEED, 1, 2, total // 2, chunks=3)
    torch.cuda.synchronize()
    assert torch.equal(ref_t[1::2][: total // 2], t_odd[: total // 2]) and torch.equal(ref_t8[1::2][: total // 2], t8_odd[: total // 2])
    gs.launch_status()
    img, img8 = gs.untile(ref_t, ref_t8, 0, 1, total)
    torch.cuda.synchronize()
    st = dict(zip(("rays", "casts", "tests", "samples"), ref_s.cpu().tolist()))
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(img.cpu().numpy(), img8.cpu().numpy(), st, mean, rgb8, ost, what=f"chunked refraction, {kind}", hdr=True,
                  abs_floor=fixed_point_floor(sc))
    gs.close()
    sc.free()


def test_the_refraction_cliff_is_gone(gpu):
    """until round 5 a launch of a glass mesh left the parked-walk kernel for the static one (3x slower) as soon as
    samples x 2^(max_depth + 2) passed 2^30 -- 4,097 spp at depth 16.  Now the launch is cut into the sample chunks its windowed
    sums need: BASELINE configs[4]'s scene with its mesh turned to glass at 3840x2160, 16,384 spp, depth 16 (two chunks
    needed) stays on pt_render_tiles_tri_queued_refr_sph -- tiles on the mesh, its outline and the floor, bit-equal for 2, 4 and
    7 chunks; without a workspace the same launch falls back, says so, and renders the same pixels to float rounding"""
    import torch
    from rt_amd import abi, scene as S
    sc = S.build_scene(5, None, None, 16384, 16)
    sc.meshes[0].flags = abi.M_REFRACTION
    gs = gpu.GpuScene(sc)
    view = S.mesh_view_tiles(sc)
    tiles = [int(view["inside"][len(view["inside"]) // 2]), int(view["silhouette"][7]), int(view["outside"][-3000])]
    total = gpu.n_tiles(sc.width, sc.height)   # (a whole 4K frame has enough tiles: the suggestion is what the sums need, no more)
    assert gs.suggest_chunks(total) == 2 and gs.suggest_chunks(total, samples=8192) == 1 and gs.suggest_chunks(total, samples=8193) == 2
    assert gs.suggest_chunks(total, samples=64, max_depth=29) == 64 and gs.suggest_chunks(total, samples=64, max_depth=30) == 1
    out = {}
    for chunks in (2, 4, 7):
        got = []
        for t in tiles:
            tl, tl8, st = gs.render_tiles(SEED, t, 1, 1, chunks=chunks)
            torch.cuda.synchronize()
            assert gs.last_launch_kernel() == "pt_render_tiles_tri_queued_refr_sph", (chunks, gs.last_launch_kernel())
            got.append((tl.clone(), tl8.clone(), st.clone()))
        out[chunks] = got
    for chunks in (4, 7):
        for a, b in zip(out[2], out[chunks]):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), chunks
    # one chunk asked for, a workspace handed over: the shim raises the count to what the sums need
    ws = torch.empty(gs.shim.rt_hip_scene_chunk_workspace_bytes(gs.handle, 1), dtype=torch.uint8, device="cuda")
    tl, tl8, st = gs.render_tiles(SEED, tiles[0], 1, 1, chunks=1, workspace=ws)
    torch.cuda.synchronize()
    assert gs.last_launch_kernel() == "pt_render_tiles_tri_queued_refr_sph"
    assert torch.equal(tl, out[2][0][0]) and torch.equal(st, out[2][0][2])
    # ... and none: the table's fit = NO row
    tl, tl8, st = gs.render_tiles(SEED, tiles[0], 1, 1)
    torch.cuda.synchronize()
    assert gs.last_launch_kernel() == "pt_render_tiles_tri_big_refr"
    assert torch.equal(st, out[2][0][2])
    a, b = tl.double().cpu().numpy(), out[2][0][0].double().cpu().numpy()
    assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max()
    gs.launch_status()
    gs.close()
    sc.free()


def test_golden_config5_wide_at_its_own_resolution(gpu):
    """tests/golden/c5_wide.npz (make_golden.py c5_wide: the reference's compiled trace_path() with its mesh scan revived):
    256 tiles = 16,384 pixels of BASELINE configs[4] at its OWN 3840x2160, 2 spp -- 64 tiles that straddle the mesh's
    outline, 64 inside it, 128 over the rest of the frame, which see the mesh only through a bounce.  Each tile rendered on
    its own; pixels, bytes, rays and tests of the subset against the fixture.  (bench.py carries the same comparison on
    1,024 tiles at 4 spp into the driver's line: configs[config 5].parity.wide.)"""
    import torch
    from rt_amd import scene as S
    fr = np.load(GOLD + "/c5_wide.npz", allow_pickle=False)
    w, h, spp, depth = [int(v) for v in fr["dims"]]
    sc = S.build_scene(5, samples=spp)
    assert (sc.width, sc.height, sc.max_depth) == (w, h, depth) == (3840, 2160, 16)
    view = S.mesh_view_tiles(sc)
    n_sil, n_in, n_out = [int(v) for v in fr["groups"]]
    tiles = fr["tiles"]
    # the fixture's tiles are what the classification gives today (the groups mean what they say)
    assert set(tiles[:n_sil].tolist()) <= set(view["silhouette"].tolist()) and set(tiles[n_sil:n_sil + n_in].tolist()) <= set(view["inside"].tolist())
    assert set(tiles[n_sil + n_in:].tolist()) <= set(view["outside"].tolist()) and len(tiles) == 256
    gs = gpu.GpuScene(sc)
    t_all = torch.empty((len(tiles), 64, 3), dtype=torch.float32, device="cuda")
    t8_all = torch.empty((len(tiles), 64, 3), dtype=torch.uint8, device="cuda")
    stats = torch.zeros(4, dtype=torch.int64, device="cuda")
    for k, t in enumerate(tiles):
        gs.render_tiles(SEED, int(t), 1, 1, t_all[k:k + 1], t8_all[k:k + 1], stats)
    torch.cuda.synchronize()
    gs.launch_status()
    assert gs.last_launch_kernel() == "pt_render_tiles_tri_queued_sph"
    st = stats.cpu().tolist()
    ost = dict(rays=int(fr["stats"][0]), tests=int(fr["stats"][1]))
    assert_parity(t_all.cpu().numpy().reshape(-1, 3), t8_all.cpu().numpy().reshape(-1, 3), dict(rays=st[0], tests=st[2]), fr["mean"],
                  fr["rgb8"], ost, what="config 5 at 3840x2160, 256 tiles x 2 spp")
    gs.close()
    sc.free()


def test_chunk_suggestions_know_the_body_a_scene_takes(gpu):
    """rt_hip_suggest_chunks_depth: the pooled kernels render a tile per workgroup (>= 20 workgroups per resident slot); the
    parked-walk kernels a tile per WAVE (>= 30 rounds of workgroups); a chunk keeps >= 128 samples, of an M_REFRACTION form
    >= 64 (tools/shard_chunks.py, profiles/r05_shard_chunks.txt -- one rank's share of config 5 at N = 8 and 4096 spp: 505 ms
    with round 4's 2 chunks, 447 with 8, ideal 418).  A whole frame keeps one chunk."""
    from rt_amd import scene as S
    c4 = gpu.GpuScene(S.build_scene(4))
    total4 = gpu.n_tiles(1920, 1080)
    assert [c4.suggest_chunks((total4 + n - 1) // n) for n in (1, 2, 4, 8)] == [1, 2, 4, 7]
    c4.close()
    sc5 = S.build_scene(5)
    c5 = gpu.GpuScene(sc5)
    assert c5.kernel_name() == "pt_render_tiles_tri_queued_sph"
    total5 = gpu.n_tiles(3840, 2160)
    assert [c5.suggest_chunks((total5 + n - 1) // n) for n in (1, 2, 4, 8)] == [1, 2, 4, 8]                  # its own 4096 spp
    assert [c5.suggest_chunks((total5 + n - 1) // n, samples=256) for n in (1, 2, 4, 8)] == [1, 2, 2, 2]   # >= 128 samples per chunk
    assert c5.suggest_chunks(total5 // 8, samples=200) == 1
    c5.close()
    sc5.meshes[0].flags = gpu.abi.M_REFRACTION                      # the glass mesh: >= 64 samples per chunk
    g5 = gpu.GpuScene(sc5)
    assert [g5.suggest_chunks((total5 + n - 1) // n, samples=256, max_depth=5) for n in (1, 2, 4, 8)] == [1, 2, 4, 4]
    g5.close()
    sc5.free()
    c3 = gpu.GpuScene(S.build_scene(3))                             # config 3's own 256 spp: two chunks at most
    assert [c3.suggest_chunks((total4 + n - 1) // n) for n in (1, 2, 4, 8)] == [1, 2, 2, 2]
    c3.close()
