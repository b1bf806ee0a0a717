"""A workgroup that cannot get a slot of a per-device pool must FAIL the launch for the caller -- not hand back a
plausible-looking frame.  Its tile reads NaN (bytes 255), which is also what a legitimate NaN sample gives under the
reference's own convention (raytracer.c:218-220: NaN -> 255), so pixel values cannot carry the message: the device's status
word does (PtLaunch.status -> rt_hip_launch_status; rt_hip_render_image checks it itself).

The pools are sized so that this cannot happen in the product (pt_pool_slots_per_xcd: occupancy x CUs per XCD + 25 %); the
development build (librt_hip_dev.so, RT_HIP_POOL_SLOTS=1) gives both pools ONE slot per XCD, so all but eight workgroups of a
launch find none.  Child processes (the switch is read by the dev library only), under a time-out: no hang either.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, SEED

pytestmark = pytest.mark.gpu

DEV_LIB = os.path.join(ROOT, "raytracer.c_amd", "csrc", "librt_hip_dev.so")

CHILD = r"""
import json, sys
sys.path[:0] = [%(pkg)r, %(tests)r]
import numpy as np, torch
from rt_amd import abi, gpu as G, scene as S
from util import glass_scene
out = {}
shim = abi.load_shim()
for name, sc in (("glass", glass_scene(96, 64, 4, 5)), ("mesh", S.build_scene(5, 96, 56, 2)), ("plain", S.build_scene(2, 96, 56, 2))):
    r = {}
    try:   # the C host's entry point: the error code, with the reason
        G.render_image_host(sc, %(seed)d)
        r["image_error"] = None
    except G.ShimError as e:
        r["image_error"] = str(e)
    gs = G.GpuScene(sc)   # the tile API: NaN tiles + the status word
    r["kernel"] = gs.kernel_name()
    total = G.n_tiles(sc.width, sc.height)
    tiles, tiles8, stats = gs.render_tiles(%(seed)d, 0, 1, total)
    torch.cuda.synchronize()
    t = tiles.cpu().numpy()
    nan_tiles = int(np.isnan(t).all(axis=(1, 2)).sum())
    r["tiles"], r["nan_tiles"] = total, nan_tiles
    r["bytes_255_in_nan_tiles"] = bool((tiles8.cpu().numpy()[np.isnan(t).all(axis=(1, 2))] == 255).all())
    flags = abi.C.c_uint32(0)
    r["status_rc"] = shim.rt_hip_launch_status(0, abi.C.byref(flags))
    r["status_flags"] = flags.value
    r["status_msg"] = shim.rt_hip_last_error().decode()
    r["status_rc_again"] = shim.rt_hip_launch_status(0, abi.C.byref(flags))   # read and cleared
    gs.close()
    out[name] = r
print(json.dumps(out))
"""


def _child(slots):
    env = dict(os.environ, RT_HIP_SHIM_PATH=DEV_LIB)
    if slots:
        env["RT_HIP_POOL_SLOTS"] = str(slots)
    code = CHILD % dict(pkg=os.path.join(ROOT, "raytracer.c_amd"), tests=os.path.join(ROOT, "tests"), seed=SEED)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_a_missing_pool_slot_fails_the_launch_instead_of_returning_nan_tiles():
    d = _child(1)
    g = d["glass"]      # pending-ray pool
    assert g["kernel"] == "pt_render_tiles_refr_pool"
    assert g["image_error"] and "failed (-4)" in g["image_error"] and "no free slot in the pending-ray pool" in g["image_error"]
    assert 0 < g["nan_tiles"] < g["tiles"] and g["bytes_255_in_nan_tiles"]
    assert g["status_rc"] == -4 and g["status_flags"] == 1 and "pending-ray pool" in g["status_msg"]
    assert g["status_rc_again"] == 0
    m = d["mesh"]       # parked-walk workspace
    assert m["kernel"] == "pt_render_tiles_tri_queued_sph"
    assert m["image_error"] and "no free slot in the parked-walk workspace" in m["image_error"]
    assert 0 < m["nan_tiles"] < m["tiles"] and m["bytes_255_in_nan_tiles"]
    assert m["status_rc"] == -4 and m["status_flags"] == 2
    p = d["plain"]      # a kernel that takes no slot is untouched
    assert p["image_error"] is None and p["nan_tiles"] == 0 and p["status_rc"] == 0 and p["status_flags"] == 0


def test_the_same_library_with_its_own_pool_sizes_reports_nothing():
    d = _child(0)
    for name in ("glass", "mesh", "plain"):
        r = d[name]
        assert r["image_error"] is None and r["nan_tiles"] == 0 and r["status_rc"] == 0 and r["status_flags"] == 0, (name, r)


def test_pool_sizes_are_derived_from_the_device_with_slack():
    """pt_pool_slots_per_xcd: CUs per XCD x the most workgroups of a slot-taking kernel a CU holds, + 25 %, a multiple of 32:
    on an MI355X (32 CUs per XCD, 4 workgroups of the parked-walk / pooled refraction kernels per CU) 160, not the 128 with
    zero slack of round 4"""
    import ctypes as C
    from rt_amd import abi
    shim = abi.load_shim()
    park, pend = C.c_uint32(0), C.c_uint32(0)
    assert shim.rt_hip_selftest_pool_slots(0, C.byref(park), C.byref(pend)) == 0
    cus = C.c_int(0)
    shim.rt_hip_device_info(0, None, 0, C.byref(cus))
    per_xcd = (cus.value + 7) // 8
    for n in (park.value, pend.value):
        assert n % 32 == 0 and per_xcd * 1.25 <= n <= per_xcd * 8 * 1.25 + 32, (n, per_xcd)
    if cus.value == 256:
        assert park.value >= 160 and pend.value >= 160


def test_the_pending_ray_pool_grows_to_a_deep_launch_and_shrinks_back():
    """one deep glass-mesh launch needs ~6.5 GB of pending-ray stacks (31 entries x 2,048 columns x 1,280 slots); the pool used
    to stay that large until rt_hip_release_cache() (round-4 advisor finding).  Now it is rebuilt to fit once 16 launches in a
    row needed at most a quarter of it -- and the frames on either side of the rebuild are the same"""
    import ctypes as C
    import torch
    from rt_amd import abi, gpu as G
    from util import convex_body_scene, glass_scene
    shim = abi.load_shim()
    shim.rt_hip_release_cache()
    park, pend = C.c_size_t(0), C.c_size_t(0)

    def pend_bytes():
        assert shim.rt_hip_pool_bytes(0, C.byref(park), C.byref(pend)) == 0
        return pend.value
    assert pend_bytes() == 0
    deep = convex_body_scene(5, 32, 20, 2)[0]
    deep.max_depth = 29
    deep.meshes[0].flags = abi.M_REFRACTION
    gs = G.GpuScene(deep)
    gs.render_image(SEED)
    assert gs.last_launch_kernel().startswith("pt_render_tiles_tri_queued_refr")
    big = pend_bytes()
    assert big > 5 << 30 and park.value > 300 << 20
    gs.close()
    small_sc = glass_scene(64, 40, 4, 5)
    gs = G.GpuScene(small_sc)
    first = gs.render_image(SEED)
    for k in range(14):
        gs.render_image(SEED)
        assert pend_bytes() == big, k          # 15 small launches so far: still the deep launch's pool
    again = gs.render_image(SEED)              # the 16th: rebuilt to fit
    small = pend_bytes()
    assert small < 1 << 30 and small * 8 < big, (small, big)
    assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1]) and first[2] == again[2]
    gs.launch_status()
    gs.close()
    shim.rt_hip_release_cache()
    assert pend_bytes() == 0
