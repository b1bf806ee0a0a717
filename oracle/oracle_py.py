"""oracle/oracle_py.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes bindings of the two CPU checkers:
  * PtOracle  -- oracle/libpt_oracle.so, our C restatement of the reference hot path;
  * RefOracle -- oracle/_ref/libref_oracle_d<N>.so, the reference's own compiled code
                 (one library per MAX_DEPTH, see oracle/Makefile).
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this.
Nothing here reads /root/reference at run time: the _ref libraries are prebuilt.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "raytracer.c_amd"))
from rt_amd import abi  # noqa: E402  (struct layouts only)

PT_PATH = os.path.join(HERE, "libpt_oracle.so")
REF_DEPTHS = (4, 5, 8, 16)
INTEGRATORS = {"path": 0, "whitted": 1}  # render()'s `#if 1` (raytracer.c:207-211): trace_path / cast_ray


def ref_path(depth):
    return os.path.join(HERE, "_ref", f"libref_oracle_d{depth}.so")


def ref_available(depth=None):
    depths = REF_DEPTHS if depth is None else (depth,)
    return all(os.path.exists(ref_path(d)) for d in depths)


def _dbl(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, ctype=C.c_double):
    return a.ctypes.data_as(C.POINTER(ctype))


def _pixels_arg(pixels, w, h):
    if pixels is None:
        return None, w * h, None
    px = np.ascontiguousarray(pixels, dtype=np.uint32)
    return _ptr(px, C.c_uint32), px.size, px


class _Common:
    """Primitive known-answer wrappers shared by both libraries (prefix differs)."""

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def init_camera(self, pos, target, w, h):
        cam = abi.Camera()
        self._fn("init_camera")(C.byref(cam), _ptr(_dbl(pos)), _ptr(_dbl(target)), C.c_int(w), C.c_int(h))
        return cam

    def camera_ray(self, cam, u, v):
        out = np.zeros(6)
        self._fn("camera_ray")(C.byref(cam), C.c_double(u), C.c_double(v), _ptr(out))
        return out

    def intersect_sphere(self, ray, center, radius):
        t = C.c_double(0)
        fn = self._fn("intersect_sphere")
        fn.restype = C.c_int
        ok = fn(_ptr(_dbl(ray)), _ptr(_dbl(center)), C.c_double(radius), C.byref(t))
        return bool(ok), t.value

    def intersect_triangle(self, ray, verts15):
        out = np.zeros(3)
        fn = self._fn("intersect_triangle")
        fn.restype = C.c_int
        ok = fn(_ptr(_dbl(ray)), _ptr(_dbl(verts15)), _ptr(out))
        return bool(ok), out

    def surface_normal(self, v9):
        out = np.zeros(3)
        self._fn("surface_normal")(_ptr(_dbl(v9)), _ptr(out))
        return out

    def reflect(self, i, n):
        out = np.zeros(3)
        self._fn("reflect")(_ptr(_dbl(i)), _ptr(_dbl(n)), _ptr(out))
        return out

    def refract(self, i, n, iot):
        out = np.zeros(3)
        self._fn("refract")(_ptr(_dbl(i)), _ptr(_dbl(n)), C.c_double(iot), _ptr(out))
        return out

    def checkered(self, color, u, v, m):
        out = np.zeros(3)
        self._fn("checkered")(_ptr(_dbl(color)), C.c_double(u), C.c_double(v), C.c_double(m), _ptr(out))
        return out

    def intersect_scene(self, ray, objs, n):
        pn, tuv, oid = np.zeros(6), np.zeros(3), C.c_uint32(0)
        fn = self._fn("intersect_scene")
        fn.restype = C.c_int
        ok = fn(_ptr(_dbl(ray)), objs, C.c_size_t(n), _ptr(pn), _ptr(tuv), C.byref(oid))
        return bool(ok), pn, tuv, oid.value

    def random_doubles(self, seed, pixel, sample, count):
        out = np.zeros(count)
        self._fn("random_doubles")(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_int(count), _ptr(out))
        return out


class PtOracle(_Common):
    prefix = "pto_"

    def __init__(self):
        if not os.path.exists(PT_PATH):
            raise RuntimeError(f"{PT_PATH} missing: run `make oracle`")
        self.lib = C.CDLL(PT_PATH)

    def render_pixels(self, scene, seed, pixels=None, spp=None, max_depth=None, want_rgb8=True,
                      integrator="path"):
        """-> mean (npix,3) float64, rgb8 (npix,3) uint8, stats dict.
        integrator: "path" = trace_path, "whitted" = cast_ray (raytracer.c:556-641)."""
        ptr, npix, keep = _pixels_arg(pixels, scene.width, scene.height)
        mean = np.zeros((npix, 3))
        rgb8 = np.zeros((npix, 3), dtype=np.uint8)
        stats = (C.c_longlong * 4)()
        self.lib.pto_render_pixels_with(
            C.c_int(INTEGRATORS[integrator]), scene.objects, C.c_size_t(scene.n_objects), scene.meshes, C.c_size_t(scene.n_meshes),
            C.byref(scene.camera), C.c_int(scene.width), C.c_int(scene.height),
            C.c_int(spp or scene.samples), C.c_int(scene.max_depth if max_depth is None else max_depth),
            C.c_uint64(seed), ptr, C.c_size_t(npix), _ptr(mean), _ptr(rgb8, C.c_uint8) if want_rgb8 else None, stats)
        return mean, rgb8, dict(rays=stats[0], tests=stats[1], casts=stats[2], draws=stats[3])

    def trace_sample(self, scene, x, y, s, seed, max_depth=None, integrator="path"):
        rgb = np.zeros(3)
        stats = (C.c_longlong * 4)()
        self.lib.pto_trace_sample_with(
            C.c_int(INTEGRATORS[integrator]), scene.objects, C.c_size_t(scene.n_objects), scene.meshes, C.c_size_t(scene.n_meshes),
            C.byref(scene.camera), C.c_int(scene.width), C.c_int(scene.height),
            C.c_int(scene.max_depth if max_depth is None else max_depth), C.c_uint32(x), C.c_uint32(y),
            C.c_uint32(s), C.c_uint64(seed), _ptr(rgb), stats)
        return rgb, dict(rays=stats[0], tests=stats[1], casts=stats[2], draws=stats[3])

    def intersect_mesh_scene(self, ray, scene):
        pn, tuv, oid, tests = np.zeros(6), np.zeros(3), C.c_uint32(0), C.c_longlong(0)
        fn = self.lib.pto_intersect_mesh_scene
        fn.restype = C.c_int
        ok = fn(_ptr(_dbl(ray)), scene.objects, C.c_size_t(scene.n_objects), scene.meshes, C.c_size_t(scene.n_meshes),
                _ptr(pn), _ptr(tuv), C.byref(oid), C.byref(tests))
        return dict(hit=bool(ok), point=pn[:3].copy(), normal=pn[3:].copy(), min_t=tuv[0], u=tuv[1], v=tuv[2],
                    id=oid.value, tests=tests.value)

    def tonemap(self, mean):
        mean = _dbl(mean)
        out = np.zeros(mean.shape, dtype=np.uint8)
        self.lib.pto_tonemap(_ptr(mean), C.c_size_t(mean.size // 3), _ptr(out, C.c_uint8))
        return out


class RefOracle(_Common):
    """The reference's compiled trace_path()/intersect() at one MAX_DEPTH.  Spheres only
    (the reference's live intersect() has no mesh branch, raytracer.c:401-412)."""
    prefix = "ref_"

    def __init__(self, depth):
        path = ref_path(depth)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing (built by `make oracle` where /root/reference exists)")
        self.lib = C.CDLL(path)
        self.depth = depth
        assert self.lib.ref_max_depth() == depth

    def layout(self):
        out = (C.c_uint64 * 16)()
        self.lib.ref_layout(out)
        return list(out)

    def render_pixels(self, scene, seed, pixels=None, spp=None, want_rgb8=True, integrator="path"):
        assert scene.n_meshes == 0, "the compiled reference scans spheres only"
        self.lib.ref_set_integrator(C.c_int(INTEGRATORS[integrator]))
        ptr, npix, keep = _pixels_arg(pixels, scene.width, scene.height)
        mean = np.zeros((npix, 3))
        rgb8 = np.zeros((npix, 3), dtype=np.uint8)
        stats = (C.c_longlong * 2)()
        self.lib.ref_render_pixels(
            scene.objects, C.c_size_t(scene.n_objects), C.byref(scene.camera), C.c_int(scene.width),
            C.c_int(scene.height), C.c_int(spp or scene.samples), C.c_uint64(seed), ptr, C.c_size_t(npix),
            _ptr(mean), _ptr(rgb8, C.c_uint8) if want_rgb8 else None, stats)
        return mean, rgb8, dict(rays=stats[0], tests=stats[1])

    def trace_sample(self, scene, x, y, s, seed, integrator="path"):
        rgb = np.zeros(3)
        stats = (C.c_longlong * 3)()
        self.lib.ref_set_integrator(C.c_int(INTEGRATORS[integrator]))
        self.lib.ref_trace_sample(
            scene.objects, C.c_size_t(scene.n_objects), C.byref(scene.camera), C.c_int(scene.width),
            C.c_int(scene.height), C.c_uint32(x), C.c_uint32(y), C.c_uint32(s), C.c_uint64(seed), _ptr(rgb), stats)
        return rgb, dict(rays=stats[0], tests=stats[1], draws=stats[2])

    def render_as_shipped(self, scene, libc_seed, spp=None, threads=1):
        """The reference's render() itself, libc rand(): -> (h,w,3) uint8, stats.
        threads=1 is its deterministic (and fastest) mode."""
        self.lib.ref_set_threads(C.c_int(threads))
        fb = np.zeros((scene.height, scene.width, 3), dtype=np.uint8)
        stats = (C.c_longlong * 2)()
        self.lib.ref_render_as_shipped(
            _ptr(fb, C.c_uint8), scene.objects, C.c_size_t(scene.n_objects), C.byref(scene.camera),
            C.c_int(scene.width), C.c_int(scene.height), C.c_int(spp or scene.samples), C.c_uint(libc_seed), stats)
        return fb, dict(rays=stats[0], tests=stats[1])

    def render_loop_libc(self, scene, libc_seed, spp=None):
        fb = np.zeros((scene.height, scene.width, 3), dtype=np.uint8)
        stats = (C.c_longlong * 2)()
        self.lib.ref_render_loop_libc(
            _ptr(fb, C.c_uint8), scene.objects, C.c_size_t(scene.n_objects), C.byref(scene.camera),
            C.c_int(scene.width), C.c_int(scene.height), C.c_int(spp or scene.samples), C.c_uint(libc_seed), stats)
        return fb, dict(rays=stats[0], tests=stats[1])


def ref_mesh_path(depth):
    return os.path.join(HERE, "_ref", f"libref_mesh_d{depth}.so")


def ref_mesh_available(depth=None):
    depths = REF_DEPTHS if depth is None else (depth,)
    return all(os.path.exists(ref_mesh_path(d)) for d in depths)


class RefMeshOracle(RefOracle):
    """The reference's compiled trace_path() / cast_ray() and primitives with its commented-out
    mesh scan (raytracer.c:417-435) revived around them (ref_harness.c, ORACLE_MESH_HOOK).
    Scenes without meshes go through the same hook and must equal RefOracle's bit for bit."""

    def __init__(self, depth):
        path = ref_mesh_path(depth)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing (built by `make oracle` where /root/reference exists)")
        self.lib = C.CDLL(path)
        self.depth = depth
        assert self.lib.ref_max_depth() == depth and self.lib.ref_mesh_hook() == 1
        self.lib.ref_mesh_layout.restype = C.c_uint64
        assert self.lib.ref_mesh_layout() == C.sizeof(abi.MeshObject)

    def render_pixels(self, scene, seed, pixels=None, spp=None, want_rgb8=True, integrator="path"):
        self.lib.ref_set_integrator(C.c_int(INTEGRATORS[integrator]))
        ptr, npix, keep = _pixels_arg(pixels, scene.width, scene.height)
        mean = np.zeros((npix, 3))
        rgb8 = np.zeros((npix, 3), dtype=np.uint8)
        stats = (C.c_longlong * 2)()
        self.lib.ref_mesh_render_pixels(
            scene.objects, C.c_size_t(scene.n_objects), scene.meshes, C.c_size_t(scene.n_meshes),
            C.byref(scene.camera), C.c_int(scene.width), C.c_int(scene.height), C.c_int(spp or scene.samples),
            C.c_uint64(seed), ptr, C.c_size_t(npix), _ptr(mean), _ptr(rgb8, C.c_uint8) if want_rgb8 else None, stats)
        return mean, rgb8, dict(rays=stats[0], tests=stats[1])

    def trace_sample(self, scene, x, y, s, seed, integrator="path"):
        rgb = np.zeros(3)
        stats = (C.c_longlong * 3)()
        self.lib.ref_set_integrator(C.c_int(INTEGRATORS[integrator]))
        self.lib.ref_mesh_trace_sample(
            scene.objects, C.c_size_t(scene.n_objects), scene.meshes, C.c_size_t(scene.n_meshes),
            C.byref(scene.camera), C.c_int(scene.width), C.c_int(scene.height), C.c_uint32(x), C.c_uint32(y),
            C.c_uint32(s), C.c_uint64(seed), _ptr(rgb), stats)
        return rgb, dict(rays=stats[0], tests=stats[1], draws=stats[2])

    def intersect_mesh_scene(self, ray, scene):
        """One call of the revived scan -> dict(hit, point, normal, t/u/v as the literal block leaves
        them (stale), min_t and the winner's own u/v, object id, primitive tests)."""
        pn, tuv, win, oid, tests = np.zeros(6), np.zeros(3), np.zeros(3), C.c_uint32(0), C.c_longlong(0)
        fn = self.lib.ref_intersect_mesh_scene
        fn.restype = C.c_int
        ok = fn(_ptr(_dbl(ray)), scene.objects, C.c_size_t(scene.n_objects), scene.meshes, C.c_size_t(scene.n_meshes),
                _ptr(pn), _ptr(tuv), _ptr(win), C.byref(oid), C.byref(tests))
        return dict(hit=bool(ok), point=pn[:3].copy(), normal=pn[3:].copy(), t_stale=tuv[0], u=tuv[1], v=tuv[2],
                    min_t=win[0], u_win=win[1], v_win=win[2], id=oid.value, tests=tests.value)
