"""Scenes and cameras for the Python-side drivers (tests, bench.py, smoke): thin
calls into libraytracer_amd.so, so Python and the C host see the same bytes.
"""
import ctypes as C
from dataclasses import dataclass, field

from . import abi


@dataclass
class Scene:
    config: int
    width: int
    height: int
    samples: int
    max_depth: int
    objects: C.Array          # Object[n_objects]
    meshes: C.Array           # MeshObject[n_meshes] (vertex arrays owned by the host library)
    n_objects: int
    n_meshes: int
    n_triangles: int
    camera: abi.Camera
    _keep: list = field(default_factory=list)

    @property
    def n_primitives(self):
        return self.n_objects + self.n_triangles

    def hip_meshes(self):
        """MeshObject[] -> RtHipMesh[] (host pointers shared, nothing copied)."""
        arr = (abi.RtHipMesh * max(self.n_meshes, 1))()
        for i in range(self.n_meshes):
            m = self.meshes[i]
            arr[i].flags = m.flags
            arr[i].color[:] = m.color.tuple()
            arr[i].emission[:] = m.emission.tuple()
            arr[i].num_triangles = m.mesh.num_triangles
            arr[i].vertices = m.mesh.vertices
        return arr

    def free(self):
        if self.n_meshes:
            abi.load_host().rt_scene_free_meshes(self.meshes, self.n_meshes)
            self.n_meshes = 0


def make_camera(width, height, pos, target):
    """init_camera() of the boundary (reference raytracer.c:47-75)."""
    host = abi.load_host()
    cam = abi.Camera()
    opt = abi.Options()
    opt.width, opt.height, opt.samples = width, height, 1
    host.init_camera(C.byref(cam), abi.Vec3(*pos), abi.Vec3(*target), C.byref(opt))
    return cam


def scene_info(config):
    host = abi.load_host()
    info = abi.RtSceneInfo()
    if host.rt_scene_info(config, C.byref(info)) != 0:
        raise ValueError(f"unknown scene config {config}")
    return info


def build_scene(config, width=None, height=None, samples=None, max_depth=None):
    """BASELINE.json configs[config-1]; size / spp / depth default to the
    configuration's nominal values and can be scaled down for tests."""
    host = abi.load_host()
    info = scene_info(config)
    w = width or info.width
    h = height or info.height
    objs = (abi.Object * max(info.n_objects, 1))()
    meshes = (abi.MeshObject * max(info.n_meshes, 1))()
    if host.rt_scene_build(config, w, h, objs, meshes) != 0:
        raise RuntimeError(f"rt_scene_build({config}) failed")
    cam = make_camera(w, h, tuple(info.cam_pos), tuple(info.cam_target))
    return Scene(config=config, width=w, height=h, samples=samples or info.samples,
                 max_depth=info.max_depth if max_depth is None else max_depth, objects=objs, meshes=meshes,
                 n_objects=info.n_objects, n_meshes=info.n_meshes, n_triangles=info.n_triangles, camera=cam)


def custom_scene(objects, width, height, samples, max_depth, cam_pos, cam_target, meshes=None):
    """A hand-made scene: objects = list of dicts(flags, radius, center, color, emission)."""
    arr = (abi.Object * max(len(objects), 1))()
    for i, o in enumerate(objects):
        arr[i].flags = o["flags"]
        arr[i].radius = o["radius"]
        arr[i].center = abi.Vec3(*o["center"])
        arr[i].color = abi.Vec3(*o["color"])
        arr[i].emission = abi.Vec3(*o.get("emission", (0, 0, 0)))
    keep = []
    marr = (abi.MeshObject * max(len(meshes or []), 1))()
    ntri = 0
    for i, m in enumerate(meshes or []):
        tris = m["triangles"]  # list of 3 x (pos xyz, tex uv) rows, flattened per vertex
        verts = (abi.Vertex * (3 * len(tris)))()
        for t, tri in enumerate(tris):
            for k in range(3):
                p = tri[k]
                verts[3 * t + k].pos = abi.Vec3(p[0], p[1], p[2])
                verts[3 * t + k].tex = abi.Vec2(p[3] if len(p) > 3 else 0.0, p[4] if len(p) > 4 else 0.0)
        keep.append(verts)
        marr[i].flags = m["flags"]
        marr[i].color = abi.Vec3(*m["color"])
        marr[i].emission = abi.Vec3(*m.get("emission", (0, 0, 0)))
        marr[i].mesh.num_triangles = len(tris)
        marr[i].mesh.vertices = C.cast(verts, C.POINTER(abi.Vertex))
        ntri += len(tris)
    cam = make_camera(width, height, cam_pos, cam_target)
    sc = Scene(config=0, width=width, height=height, samples=samples, max_depth=max_depth, objects=arr,
               meshes=marr, n_objects=len(objects), n_meshes=len(meshes or []), n_triangles=ntri, camera=cam,
               _keep=keep)
    sc.free = lambda: None  # vertex arrays are Python-owned
    return sc
