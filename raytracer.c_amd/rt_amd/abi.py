"""ctypes mirror of the C boundary: include/raytracer.h (the reference's structs,
reference raytracer.h:60-131), include/rt_hip.h (the HIP shim's C-ABI) and
raytracer.c_amd/host/scenes.h.  Plumbing only: no arithmetic happens here.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # raytracer.c_amd/
REPO_ROOT = os.path.dirname(PKG_DIR)
SHIM_PATH = os.environ.get("RT_HIP_SHIM_PATH") or os.path.join(PKG_DIR, "csrc", "librt_hip.so")  # env: dev builds
HOST_PATH = os.path.join(PKG_DIR, "host", "libraytracer_amd.so")

M_DEFAULT, M_REFLECTION, M_REFRACTION, M_CHECKERED = 2, 4, 8, 16
TILE, TILE_PIXELS, TILE_FLOATS = 8, 64, 192
STAT_RAYS, STAT_CASTS, STAT_TESTS, STAT_SAMPLES, NSTATS = 0, 1, 2, 3, 4
FAIL_ALLOC_PARK_WS, FAIL_ALLOC_WIDE_PEND = 1, 2   # rt_hip_selftest_fail_alloc
FAIL_PEND_SLOT, FAIL_PARK_SLOT = 1, 2             # rt_hip_launch_status


class Vec2(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double)]


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Object(C.Structure):  # reference raytracer.h:104-111 == RtHipSphere, 88 B
    _fields_ = [("flags", C.c_uint32), ("radius", C.c_double), ("center", Vec3), ("color", Vec3),
                ("emission", Vec3)]


class Vertex(C.Structure):  # reference raytracer.h:61 == RtHipVertex, 40 B
    _fields_ = [("pos", Vec3), ("tex", Vec2)]


class TriangleMesh(C.Structure):  # reference raytracer.h:77-81
    _fields_ = [("num_triangles", C.c_size_t), ("vertices", C.POINTER(Vertex))]


class MeshObject(C.Structure):  # include/raytracer.h extension
    _fields_ = [("flags", C.c_uint32), ("color", Vec3), ("emission", Vec3), ("mesh", TriangleMesh)]


class Camera(C.Structure):  # reference raytracer.h:121-124 == RtHipCamera, 96 B
    _fields_ = [("position", Vec3), ("horizontal", Vec3), ("vertical", Vec3), ("lower_left_corner", Vec3)]


class Options(C.Structure):  # reference raytracer.h:126-131, 56 B
    _fields_ = [("background", Vec3), ("result", C.c_char_p), ("obj", C.c_char_p), ("width", C.c_int),
                ("height", C.c_int), ("samples", C.c_int)]


class Ray(C.Structure):
    _fields_ = [("origin", Vec3), ("direction", Vec3)]


class Hit(C.Structure):  # reference raytracer.h:113-119, 80 B
    _fields_ = [("t", C.c_double), ("u", C.c_double), ("v", C.c_double), ("point", Vec3), ("normal", Vec3),
                ("object_id", C.c_uint32)]


class RtHipMesh(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("color", C.c_double * 3), ("emission", C.c_double * 3),
                ("num_triangles", C.c_size_t), ("vertices", C.POINTER(Vertex))]


class RtHipParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint64), ("tile_first", C.c_uint32), ("tile_stride", C.c_uint32),
                ("tile_count", C.c_uint32), ("integrator", C.c_uint32)]


TRACE_PATH, CAST_RAY = 0, 1  # RtHipParams.integrator / rt_set_integrator()
INTEGRATORS = {"path": TRACE_PATH, "whitted": CAST_RAY}


class RtSceneInfo(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("samples", C.c_int), ("max_depth", C.c_int),
                ("cam_pos", C.c_double * 3), ("cam_target", C.c_double * 3), ("n_objects", C.c_size_t),
                ("n_meshes", C.c_size_t), ("n_triangles", C.c_size_t)]


# every symbol include/rt_hip.h declares: name -> (restype, argtypes)
SHIM_SYMBOLS = {
    "rt_hip_device_count": (C.c_int, []),
    "rt_hip_last_error": (C.c_char_p, []),
    "rt_hip_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "rt_hip_scene_create": (C.c_int, [C.POINTER(Object), C.c_size_t, C.POINTER(RtHipMesh), C.c_size_t, C.c_int,
                                      C.POINTER(C.c_void_p)]),
    "rt_hip_scene_destroy": (None, [C.c_void_p]),
    "rt_hip_scene_device": (C.c_int, [C.c_void_p]),
    "rt_hip_scene_primitives": (C.c_size_t, [C.c_void_p]),
    "rt_hip_scene_hull_facets": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_hip_kernel_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "rt_hip_render_tiles": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RtHipParams), C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_hip_chunk_workspace_bytes": (C.c_size_t, [C.c_uint32]),
    "rt_hip_suggest_chunks": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_int32]),
    "rt_hip_suggest_chunks_depth": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]),
    "rt_hip_scene_chunk_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32]),
    "rt_hip_render_tiles_chunked": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RtHipParams), C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_hip_selftest_math": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "rt_hip_selftest_xcc": (C.c_int, [C.c_uint32, C.c_void_p, C.c_int]),
    "rt_hip_selftest_intersect": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int]),
    "rt_hip_untile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_hip_set_cancel_flag": (None, [C.c_void_p]),
    "rt_hip_release_cache": (None, []),
    "rt_hip_cache_builds": (C.c_uint64, []),
    "rt_hip_set_device_map": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "rt_hip_last_image_phases": (None, [C.POINTER(C.c_double)]),
    "rt_hip_pool_bytes": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rt_hip_last_launch_kernel": (C.c_char_p, []),
    "rt_hip_kernel_count": (C.c_int, []),
    "rt_hip_kernel_launches": (C.c_char_p, [C.c_int, C.POINTER(C.c_uint64)]),
    "rt_hip_kernel_for_class": (C.c_char_p, [C.c_void_p]),
    "rt_hip_launch_status": (C.c_int, [C.c_int, C.POINTER(C.c_uint32)]),
    "rt_hip_selftest_fail_alloc": (None, [C.c_uint32]),
    "rt_hip_selftest_pool_slots": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_hip_render_image": (C.c_int, [C.POINTER(Object), C.c_size_t, C.POINTER(RtHipMesh), C.c_size_t,
                                      C.POINTER(Camera), C.POINTER(RtHipParams), C.c_int, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
}

# every function include/raytracer.h declares (+ scenes.h), same idea
HOST_SYMBOLS = {
    "random_double": (C.c_double, []),
    "random_range": (C.c_double, [C.c_double, C.c_double]),
    "point_at": (Vec3, [C.POINTER(Ray), C.c_double]),
    "calculate_surface_normal": (Vec3, [Vec3, Vec3, Vec3]),
    "clamp": (Vec3, [Vec3]),
    "intersect_sphere": (C.c_bool, [C.POINTER(Ray), Vec3, C.c_double, C.POINTER(Hit)]),
    "intersect_triangle": (C.c_bool, [C.POINTER(Ray), Vertex, Vertex, Vertex, C.POINTER(Hit)]),
    "print_v": (None, [C.c_char_p, Vec3]),
    "print_m": (None, [C.POINTER(C.c_double)]),
    "init_camera": (None, [C.POINTER(Camera), Vec3, Vec3, C.POINTER(Options)]),
    "render": (None, [C.c_void_p, C.POINTER(Object), C.c_size_t, C.POINTER(Camera), C.POINTER(Options)]),
    "load_obj": (C.c_bool, [C.c_char_p, C.POINTER(TriangleMesh)]),
    "render_ex": (None, [C.c_void_p, C.c_void_p, C.POINTER(Object), C.c_size_t, C.POINTER(MeshObject), C.c_size_t,
                         C.POINTER(Camera), C.POINTER(Options)]),
    "rt_set_max_depth": (None, [C.c_int]),
    "rt_set_seed": (None, [C.c_uint64]),
    "rt_set_integrator": (None, [C.c_int]),
    "rt_get_integrator": (C.c_int, []),
    "rt_set_devices": (None, [C.c_int]),
    "rt_get_max_depth": (C.c_int, []),
    "rt_get_seed": (C.c_uint64, []),
    "rt_set_cancel_flag": (None, [C.c_void_p]),
    "rt_last_render_cancelled": (C.c_int, []),
    "rt_last_render_seconds": (C.c_double, []),
    "rt_last_ray_bounces": (C.c_longlong, []),
    "rt_scene_info": (C.c_int, [C.c_int, C.POINTER(RtSceneInfo)]),
    "rt_scene_build": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(Object), C.POINTER(MeshObject)]),
    "rt_scene_free_meshes": (None, [C.POINTER(MeshObject), C.c_size_t]),
    "rt_mesh_flip_winding": (None, [C.POINTER(TriangleMesh)]),
    "stbi_write_png": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
}
HOST_DATA = ["ray_count", "intersection_test_count"]


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_shim = None
_host = None


def load_shim():
    """librt_hip.so.  Raises if it is not built: there is no fallback path."""
    global _shim
    if _shim is None:
        # PyTorch-ROCm bundles its own libamdhip64.so.7 / librccl.so.1 / libhsa-runtime64.so.1.
        # The dynamic loader de-duplicates by SONAME only for libraries loaded EARLIER, so
        # torch must come first: the shim then binds to the very runtime instance torch uses
        # (one HIP runtime per process => torch streams and device pointers are valid in the
        # shim).  Loading the shim first would pull in /opt/rocm's copies next to torch's.
        import torch  # noqa: F401
        if not os.path.exists(SHIM_PATH):
            raise RuntimeError(f"{SHIM_PATH} is missing: build it with `make shim` "
                               "(or __graft_entry__.build()); this package has no CPU fallback")
        _shim = _bind(C.CDLL(SHIM_PATH, mode=C.RTLD_GLOBAL), SHIM_SYMBOLS)
    return _shim


def load_host():
    """libraytracer_amd.so: the reference API + scene builders."""
    global _host
    if _host is None:
        load_shim()
        if not os.path.exists(HOST_PATH):
            raise RuntimeError(f"{HOST_PATH} is missing: build it with `make host`")
        _host = _bind(C.CDLL(HOST_PATH), HOST_SYMBOLS)
    return _host
