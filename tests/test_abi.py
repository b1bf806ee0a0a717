"""The C-ABI boundary without a GPU: the shared libraries load, export every symbol the
headers declare, keep the reference's struct layouts, and FAIL LOUDLY (no CPU fallback)
when no device is present."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, pattern):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(pattern, text)))


def test_shim_exports_every_declared_symbol():
    from rt_amd import abi
    names = _declared("rt_hip.h", r"\b(rt_hip_[a-z_]+)\s*\(")
    assert len(names) >= 10
    assert sorted(abi.SHIM_SYMBOLS) == names, "abi.py and rt_hip.h disagree"
    lib = C.CDLL(abi.SHIM_PATH)
    for n in names:
        assert getattr(lib, n) is not None
    abi.load_shim()


def test_host_exports_reference_api():
    """every function the reference's raytracer.o exports + the two counters"""
    from rt_amd import abi
    host = abi.load_host()
    for n in ["render", "init_camera", "intersect_sphere", "intersect_triangle", "calculate_surface_normal",
              "point_at", "random_double", "random_range", "clamp", "print_v", "print_m", "load_obj"]:
        assert getattr(host, n) is not None
    for n in abi.HOST_DATA:
        C.c_longlong.in_dll(host, n)
    declared = _declared("raytracer.h", r"\b([a-z_]+)\s*\([^;{]*\)\s*;")
    for n in declared:
        assert hasattr(host, n), f"raytracer.h declares {n} but libraytracer_amd.so does not export it"


def test_struct_sizes():
    from rt_amd import abi
    assert C.sizeof(abi.Object) == 88 and C.sizeof(abi.Camera) == 96 and C.sizeof(abi.Options) == 56
    assert C.sizeof(abi.Vertex) == 40 and C.sizeof(abi.Hit) == 80 and C.sizeof(abi.Ray) == 48
    assert C.sizeof(abi.RtHipParams) == 40
    assert abi.Object.radius.offset == 8 and abi.Object.center.offset == 16
    assert abi.Object.color.offset == 40 and abi.Object.emission.offset == 64


def _no_gpu():
    from rt_amd import abi
    return abi.load_shim().rt_hip_device_count() == 0


def test_no_device_is_an_error_not_a_fallback():
    from rt_amd import abi, scene as S
    if not _no_gpu():
        pytest.skip("a GPU is visible")
    shim = abi.load_shim()
    sc = S.build_scene(1, 16, 16, 1)
    handle = C.c_void_p()
    rc = shim.rt_hip_scene_create(sc.objects, sc.n_objects, None, 0, 0, C.byref(handle))
    assert rc == -1 and not handle.value  # RT_HIP_ENODEV
    assert b"no HIP device" in shim.rt_hip_last_error()
    p = abi.RtHipParams()
    p.width, p.height, p.samples, p.max_depth = 16, 16, 1, 4
    rc = shim.rt_hip_render_image(sc.objects, sc.n_objects, None, 0, C.byref(sc.camera), C.byref(p), 1, None, None,
                                  None, None)
    assert rc == -1


def test_render_without_gpu_exits_loudly():
    """render() returns void; the reference's failure convention is stderr + EXIT_FAILURE"""
    if not _no_gpu():
        pytest.skip("a GPU is visible")
    code = ("import sys, ctypes as C; sys.path.insert(0, %r);"
            "from rt_amd import abi, scene as S; import numpy as np;"
            "sc = S.build_scene(1, 16, 16, 1); host = abi.load_host();"
            "fb = np.zeros(16*16*3, dtype=np.uint8); opt = abi.Options(); opt.width = opt.height = 16; opt.samples = 1;"
            "host.render(fb.ctypes.data, sc.objects, sc.n_objects, C.byref(sc.camera), C.byref(opt)); print('RETURNED')"
            ) % os.path.join(ROOT, "raytracer.c_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 1 and "RETURNED" not in r.stdout
    assert "GPU path failed" in r.stderr


def test_bad_arguments_rejected():
    from rt_amd import abi
    shim = abi.load_shim()
    assert shim.rt_hip_scene_create(None, 3, None, 0, 0, C.byref(C.c_void_p())) == -2  # EINVAL
    assert shim.rt_hip_untile(None, None, 0, 0, 0, 1, 1, None, None, None) == -2
    assert shim.rt_hip_render_tiles(None, None, None, None, None, None, None) == -2


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under raytracer.c_amd/ or include/ may name it"""
    bad = []
    for base in ("raytracer.c_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".pyc")) or f == "raytracer":
                    continue
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"#include\s*[<\"][^>\"]*oracle|import\s+oracle|from\s+oracle|oracle_py|libpt_oracle|"
                             r"libref_oracle|\bpto_\w+\s*\(|\bref_\w+\s*\(", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
