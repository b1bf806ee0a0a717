"""rt_hip_render_image() with n_devices > 1 -- the C host's multi-device route (north_star: "a C host ... partition the
image across the 8 GPUs ... RCCL gather"; what the reference's main.c:429 reaches through render(), replacing the
`omp parallel for` of raytracer.c:184-185) -- EXECUTED on the one GPU a test box has.

rt_hip_set_device_map({0, 0, ...}) / RT_HIP_DEVICE_MAP=0,0,... maps G logical devices onto the physical one.  Every
logical device keeps its own scene, stream, tile buffer and counters; the interleaved partition (tile_first = g + k0 G,
stride G), the slabs, the gather's first_slot[] arithmetic, the per-segment scatter and the counter sums are the very
lines that run with G distinct GPUs.  The only difference is the transport of a segment whose sender shares the root's
device: a device-to-device copy instead of ncclSend / ncclRecv -- and RT_HIP_FORCE_COMM=1 sends those through RCCL too
(one rank, sends to self).  The bar everywhere: the float frame, the bytes and all four counters are BIT-EQUAL to G = 1,
and G = 1 itself is compared with the oracle.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SEED
from util import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    from rt_amd import abi, gpu as G
    assert abi.load_shim().rt_hip_device_count() >= 1, "no HIP device: the GPU tests must run on the GPU box"
    assert torch.cuda.is_available()
    yield G
    shim = abi.load_shim()
    shim.rt_hip_set_device_map(None, 0)
    shim.rt_hip_release_cache()


def _set_map(n):
    from rt_amd import abi
    shim = abi.load_shim()
    arr = (C.c_int * max(n, 1))(*([0] * n))
    rc = shim.rt_hip_set_device_map(arr if n else None, n)
    assert rc == 0, shim.rt_hip_last_error()
    return shim


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def _scene(which):
    from rt_amd import abi, scene as S
    if which == "config2":      # 800 x 600 = 7,500 tiles: 7500 mod 8 = 4 (ragged over 8 devices), full size, few samples
        return S.build_scene(2, 800, 600, 4)
    if which == "config3":      # triangles + spheres (pt_render_tiles_tri), edge tiles on both axes
        return S.build_scene(3, 204, 116, 6)
    if which == "tiles257":     # 257 tiles in one row (a prime): every G leaves devices with unequal counts; edge tile 2 px wide
        return S.build_scene(4, 2050, 5, 8, max_depth=6)
    if which == "tiles4":       # fewer tiles than devices at G = 8: devices 4..7 hold nothing
        return S.build_scene(1, 16, 16, 4)
    if which == "glass":        # M_REFRACTION: the logical devices' kernels take slots of ONE pending-ray pool concurrently
        from util import glass_scene
        return glass_scene(120, 72, 6, 6)
    if which == "glass_mesh":   # ... and of one parked-walk workspace; 12 spp at depth 27: three sample chunks per tile (windowed sums)
        sc = S.build_scene(5, 40, 24, 12, 27)
        sc.meshes[0].flags = abi.M_REFRACTION
        return sc
    raise KeyError(which)


@pytest.mark.parametrize("which", ["config2", "config3", "tiles257", "tiles4", "glass", "glass_mesh"])
def test_logical_devices_give_the_one_device_frame_bit_for_bit(gpu, pt, which):
    from util import fixed_point_floor
    sc = _scene(which)
    _set_map(0)
    one = gpu.render_image_host(sc, SEED, n_devices=1)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert_parity(one[0], one[1], one[2], mean, rgb8, ost, what=f"{which} G=1", hdr=which.startswith("glass"),
                  abs_floor=fixed_point_floor(sc))
    if which == "glass_mesh":
        assert gpu.abi.load_shim().rt_hip_last_launch_kernel() == b"pt_render_tiles_tri_queued_refr_sph"
    _set_map(8)
    for G in (2, 3, 8):
        got = gpu.render_image_host(sc, SEED, n_devices=G)
        assert _same(got, one), (which, G, got[2], one[2])
        assert got[3] > 0
    sc.free()


def test_more_devices_than_the_map_holds_is_refused(gpu):
    from rt_amd import scene as S
    sc = S.build_scene(1, 32, 24, 2)
    _set_map(3)
    with pytest.raises(gpu.ShimError, match="asked for 4 devices, 3 available in the device map"):
        gpu.render_image_host(sc, SEED, n_devices=4)
    _set_map(0)
    have = gpu.abi.load_shim().rt_hip_device_count()
    with pytest.raises(gpu.ShimError, match=f"asked for {have + 1} devices"):
        gpu.render_image_host(sc, SEED, n_devices=have + 1)
    shim = gpu.abi.load_shim()
    bad = (C.c_int * 2)(0, have)
    assert shim.rt_hip_set_device_map(bad, 2) != 0 and b"device map entry 1" in shim.rt_hip_last_error()


def test_context_is_reused_across_frames_and_rebuilt_when_the_devices_change(gpu):
    from rt_amd import scene as S
    sc = S.build_scene(2, 120, 72, 4)
    shim = _set_map(4)
    shim.rt_hip_release_cache()
    b0 = shim.rt_hip_cache_builds()
    a = gpu.render_image_host(sc, SEED, n_devices=3)
    b = gpu.render_image_host(sc, SEED, n_devices=3)
    assert shim.rt_hip_cache_builds() == b0 + 1 and _same(a, b), "second frame on the same devices: nothing rebuilt"
    c = gpu.render_image_host(sc, SEED + 7, n_devices=3)
    assert shim.rt_hip_cache_builds() == b0 + 1 and not np.array_equal(a[0], c[0])
    d = gpu.render_image_host(sc, SEED, n_devices=2)
    assert shim.rt_hip_cache_builds() == b0 + 2 and _same(a, d), "another device count: rebuilt, same frame"
    e = gpu.render_image_host(sc, SEED, n_devices=3)
    assert shim.rt_hip_cache_builds() == b0 + 3 and _same(a, e)
    _set_map(0)
    f = gpu.render_image_host(sc, SEED, n_devices=1)
    assert _same(a, f)


def test_slabs_and_cancel_at_three_devices_return_the_finished_part(gpu):
    """a long frame on 3 logical devices is cut into slabs of EACH device's tile list; with the flag raised the render
    stops after the first slab and the gathered part equals the full frame there"""
    from rt_amd import abi, scene as S
    host = abi.load_host()
    _set_map(3)
    sc = S.build_scene(4, 1920, 1080, 100)   # 2.07e8 pixel-samples: above the slab threshold (4 slabs)
    opt = abi.Options()
    opt.width, opt.height, opt.samples = sc.width, sc.height, sc.samples
    host.rt_set_max_depth(4)
    host.rt_set_seed(SEED)
    host.rt_set_devices(1)
    one = np.zeros((sc.height, sc.width, 3), dtype=np.uint8)
    host.render(one.ctypes.data, sc.objects, sc.n_objects, C.byref(sc.camera), C.byref(opt))
    host.rt_set_devices(3)
    flag = C.c_int(0)
    host.rt_set_cancel_flag(C.byref(flag))   # registered, not raised: the slabbed path, complete
    full = np.zeros_like(one)
    host.render(full.ctypes.data, sc.objects, sc.n_objects, C.byref(sc.camera), C.byref(opt))
    assert host.rt_last_render_cancelled() == 0 and np.array_equal(full, one)
    flag.value = 1
    part = np.zeros_like(one)
    host.render(part.ctypes.data, sc.objects, sc.n_objects, C.byref(sc.camera), C.byref(opt))
    host.rt_set_cancel_flag(None)
    host.rt_set_devices(1)
    assert host.rt_last_render_cancelled() == 1
    done = part.any(axis=2)
    assert 0.15 < done.mean() < 0.35, done.mean()           # one slab of four
    assert np.array_equal(part[done], one[done])
    # each device finished the first quarter of ITS list: tile t = g + 3 k with k < count_g / 4
    tx, n_tiles = 240, 240 * 135
    tile_done = done.reshape(135, 8, 240, 8).any(axis=(1, 3)).reshape(-1)
    for g in range(3):
        count = (n_tiles - g + 2) // 3
        k1 = count // 4
        mine = tile_done[g::3]
        assert mine[k1:].sum() == 0, g
        assert mine[:k1].mean() > 0.9, g                     # (a black tile of the scene would read as not done)
    sc.free()


def test_cli_with_three_logical_devices(gpu, pt, tmp_path):
    from rt_amd import abi, scene as S
    from util import decode_png_rgb8
    exe = os.path.join(abi.PKG_DIR, "host", "raytracer")
    outs = []
    for g, env in ((1, {}), (3, {"RT_HIP_DEVICE_MAP": "0,0,0"})):
        out = str(tmp_path / f"cli_g{g}.png")
        r = subprocess.run([exe, "-w", "200", "-h", "120", "-s", "4", "-o", out, "-c", "2", "-d", "8", "-g", str(g)],
                           capture_output=True, text=True, timeout=180, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        rays = int([ln for ln in r.stdout.splitlines() if ln.startswith("cast ")][0].split()[1])
        tests = int([ln for ln in r.stdout.splitlines() if ln.startswith("checked ")][0].split()[1])
        assert f"on {g} GPU" in r.stdout
        outs.append((decode_png_rgb8(out), rays, tests))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]
    sc = S.build_scene(2, 200, 120, 4)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert (outs[1][1], outs[1][2]) == (ost["rays"], ost["tests"])
    assert np.abs(outs[1][0].reshape(-1, 3).astype(np.int16) - rgb8.astype(np.int16)).max() <= 1
    # -g beyond the map: the library's failure convention (message on stderr, EXIT_FAILURE)
    r = subprocess.run([exe, "-w", "64", "-h", "40", "-s", "2", "-o", str(tmp_path / "x.png"), "-c", "1", "-g", "4"],
                       capture_output=True, text=True, timeout=120, env=dict(os.environ, RT_HIP_DEVICE_MAP="0,0,0"))
    assert r.returncode != 0 and "asked for 4 devices, 3 available" in r.stderr


def test_reference_main_on_two_logical_devices(gpu, pt, tmp_path):
    """the reference's main.c, unmodified (oracle/_ref/ref_main_dropin), never calls rt_set_devices: RT_DEVICES=2 hands
    render() two devices, RT_HIP_DEVICE_MAP=0,0 puts both on this box's GPU"""
    from rt_amd import abi, scene as S
    from util import decode_png_rgb8
    exe = os.path.join(abi.REPO_ROOT, "oracle", "_ref", "ref_main_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_main_dropin not built (needs /root/reference at build time)")
    out = str(tmp_path / "ref_main_g2.png")
    r = subprocess.run([exe, "-w", "96", "-h", "54", "-s", "8", "-o", out], capture_output=True, text=True, timeout=180,
                       env=dict(os.environ, RT_DEVICES="2", RT_HIP_DEVICE_MAP="0,0"))
    assert r.returncode == 0, r.stderr
    img = decode_png_rgb8(out).reshape(-1, 3)
    sc = S.build_scene(4, 96, 54, 8, max_depth=5)
    mean, rgb8, ost = pt.render_pixels(sc, SEED)
    assert np.abs(img.astype(np.int16) - rgb8.astype(np.int16)).max() <= 1
    rays = int([ln for ln in r.stdout.splitlines() if ln.startswith("cast ")][0].split()[1])
    tests = int([ln for ln in r.stdout.splitlines() if ln.startswith("checked ")][0].split()[1])
    assert (rays, tests) == (ost["rays"], ost["tests"])
    # and without the map the same request must fail loudly on a one-GPU box (no silent fallback to one device)
    if abi.load_shim().rt_hip_device_count() == 1:
        r = subprocess.run([exe, "-w", "32", "-h", "24", "-s", "1", "-o", str(tmp_path / "y.png")], capture_output=True,
                           text=True, timeout=120, env=dict({k: v for k, v in os.environ.items() if k != "RT_HIP_DEVICE_MAP"},
                                                            RT_DEVICES="2"))
        assert r.returncode != 0 and "asked for 2 devices, 1 available" in r.stderr


def test_every_segment_through_rccl_with_three_logical_devices(gpu):
    """RT_HIP_FORCE_COMM=1 under the map: one communicator (one physical device = one rank), and all three segments --
    the root's own included -- travel by grouped ncclSend / ncclRecv to self into first_slot[]-addressed storage.  In a
    child process, so that a bootstrap problem or a hang is the child's, under the parent's time-out."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), "shim_map"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["frame_equal"] and d["devices"] == [1, 2, 3, 8]
    assert d["context_builds"] == 4
