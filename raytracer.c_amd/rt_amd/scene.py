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


def mesh_view_tiles(sc):
    """Classify the frame's 8x8 tiles by how the camera sees the scene's triangles' bounding ball: -> dict of tile-id arrays
    `silhouette` (the four corner pixels' un-jittered camera rays, get_camera_ray raytracer.c:375-384, disagree about hitting
    the ball: the tile straddles the mesh's outline), `inside` (all four hit), `outside` (none hits: such a tile can see the
    mesh only through a bounce).  For picking parity samples where the resolution-dependent rules of the hierarchy kernels
    (tile cones against the ball, the probe) decide differently inside one tile."""
    import numpy as np
    v = []
    for m in range(sc.n_meshes):
        n = 3 * sc.meshes[m].mesh.num_triangles
        a = np.ctypeslib.as_array(C.cast(sc.meshes[m].mesh.vertices, C.POINTER(C.c_double)), shape=(n, 5))
        v.append(a[:, :3].copy())
    v = np.concatenate(v)
    centre = 0.5 * (v.min(axis=0) + v.max(axis=0))
    radius = float(np.sqrt(((v - centre) ** 2).sum(axis=1).max()))
    cam = sc.camera
    pos = np.array(cam.position.tuple())
    hor, ver, llc = np.array(cam.horizontal.tuple()), np.array(cam.vertical.tuple()), np.array(cam.lower_left_corner.tuple())
    w, h = sc.width, sc.height
    tx, ty = (w + 7) // 8, (h + 7) // 8
    xs = np.minimum(np.arange(tx + 1) * 8, w - 1)
    ys = np.minimum(np.arange(ty + 1) * 8, h - 1)
    u = xs / (w - 1.0)
    vv = ys / (h - 1.0)
    # dir = normalize(pos - (llc + H u + V v)), origin = pos
    d = pos[None, None, :] - (llc[None, None, :] + hor[None, None, :] * u[None, :, None] + ver[None, None, :] * vv[:, None, None])
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    L = centre - pos
    tca = (d * L).sum(axis=2)
    d2 = (L * L).sum() - tca * tca
    hit = (tca > 0) & (d2 <= radius * radius)          # at the (ty + 1) x (tx + 1) grid corners
    n_hit = hit[:-1, :-1].astype(int) + hit[1:, :-1] + hit[:-1, 1:] + hit[1:, 1:]
    ids = np.arange(tx * ty, dtype=np.uint32).reshape(ty, tx)
    return dict(silhouette=ids[(n_hit > 0) & (n_hit < 4)], inside=ids[n_hit == 4], outside=ids[n_hit == 0],
                ball=(tuple(centre), radius))


def pick_evenly(ids, n):
    """n of the ids, evenly spaced through their order"""
    import numpy as np
    ids = np.asarray(ids)
    if len(ids) <= n:
        return ids.copy()
    return ids[(np.arange(n) * len(ids)) // n]
