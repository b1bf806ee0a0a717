"""Python driver of the HIP shim (include/rt_hip.h).  PyTorch is used for what it is
good at here -- device buffers, streams, torch.distributed -- and nothing else: every
pixel is computed by librt_hip.so.  There is no CPU / eager fallback: if the shim is
missing or no GPU is visible, calls raise.
"""
import ctypes as C

import torch

from . import abi
from .dist import n_tiles, rank_tiles  # noqa: F401  (re-exported)


class ShimError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        msg = abi.load_shim().rt_hip_last_error()
        raise ShimError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


class GpuScene:
    """A scene resident in HBM (rt_hip_scene_create)."""

    def __init__(self, scene, device=0):
        self.shim = abi.load_shim()
        if self.shim.rt_hip_device_count() < 1:
            raise ShimError("no HIP device visible; this package has no CPU fallback")
        self.scene = scene
        self.device = device
        self._meshes = scene.hip_meshes()
        handle = C.c_void_p()
        _check(self.shim.rt_hip_scene_create(scene.objects, scene.n_objects, self._meshes, scene.n_meshes, device,
                                             C.byref(handle)), "rt_hip_scene_create")
        self.handle = handle

    def close(self):
        if self.handle:
            self.shim.rt_hip_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def params(self, seed, first, stride, count, samples=None, max_depth=None, integrator="path"):
        p = abi.RtHipParams()
        p.integrator = abi.INTEGRATORS[integrator]
        p.width, p.height = self.scene.width, self.scene.height
        p.samples = samples or self.scene.samples
        p.max_depth = self.scene.max_depth if max_depth is None else max_depth
        p.seed = seed
        p.tile_first, p.tile_stride, p.tile_count = first, stride, count
        return p

    def kernel_name(self, integrator="path"):
        return self.shim.rt_hip_kernel_name(self.handle, abi.INTEGRATORS[integrator]).decode()

    def last_launch_kernel(self):
        """the kernel this thread's last render_tiles() launched (the launch's own facts can name another than kernel_name())"""
        return self.shim.rt_hip_last_launch_kernel().decode()

    def launch_status(self):
        """device-side failures of render launches since the last call (0 = none); raises ShimError when any"""
        flags = C.c_uint32(0)
        _check(self.shim.rt_hip_launch_status(self.device, C.byref(flags)), "rt_hip_launch_status")
        return flags.value

    def hull_facets(self):
        """(triangles marked as hull facets with the stored normal pointing outward, ... inward)"""
        plus, minus = C.c_uint32(0), C.c_uint32(0)
        _check(self.shim.rt_hip_scene_hull_facets(self.handle, C.byref(plus), C.byref(minus)), "rt_hip_scene_hull_facets")
        return plus.value, minus.value

    def suggest_chunks(self, count, samples=None, max_depth=None):
        return int(self.shim.rt_hip_suggest_chunks_depth(self.handle, count, samples or self.scene.samples,
                                                         self.scene.max_depth if max_depth is None else max_depth))

    def render_tiles(self, seed, first, stride, count, tiles=None, tiles8=None, stats=None, samples=None,
                     max_depth=None, chunks=1, workspace=None, integrator="path", camera=None):
        """Asynchronous on torch's current stream.  Returns (tiles f32 [count,64,3],
        tiles8 u8 [count,64,3], stats i64 [4]); pass buffers to reuse them.  chunks > 1 splits
        every tile's samples over that many workgroups (same image, bit for bit).
        integrator: "path" = trace_path (what the reference ships), "whitted" = cast_ray.
        camera: an abi.Camera to render with instead of the scene's own."""
        dev = torch.device("cuda", self.device)
        if tiles is None:
            tiles = torch.empty((max(count, 1), abi.TILE_PIXELS, 3), dtype=torch.float32, device=dev)
        if tiles8 is None:
            tiles8 = torch.empty((max(count, 1), abi.TILE_PIXELS, 3), dtype=torch.uint8, device=dev)
        if stats is None:
            stats = torch.zeros(abi.NSTATS, dtype=torch.int64, device=dev)
        p = self.params(seed, first, stride, count, samples, max_depth, integrator)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if chunks > 1 and workspace is None:
            workspace = torch.empty(self.shim.rt_hip_scene_chunk_workspace_bytes(self.handle, max(count, 1)), dtype=torch.uint8, device=dev)
        self._workspace = workspace  # keep alive until the stream has used it
        _check(self.shim.rt_hip_render_tiles_chunked(self.handle, C.byref(camera if camera is not None else self.scene.camera),
                                                     C.byref(p), chunks,
                                                     workspace.data_ptr() if workspace is not None else None,
                                                     tiles.data_ptr(), tiles8.data_ptr(), stats.data_ptr(),
                                                     C.c_void_p(stream)),
               "rt_hip_render_tiles_chunked")
        return tiles, tiles8, stats

    def untile(self, tiles, tiles8, first, stride, count, image=None, image8=None):
        """Scatter a compact tile buffer into row-major images on torch's current stream."""
        dev = tiles.device
        w, h = self.scene.width, self.scene.height
        if image is None:
            image = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        if image8 is None and tiles8 is not None:
            image8 = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _check(self.shim.rt_hip_untile(tiles.data_ptr(), tiles8.data_ptr() if tiles8 is not None else None, w, h,
                                       first, stride, count, image.data_ptr(),
                                       image8.data_ptr() if image8 is not None else None, C.c_void_p(stream)),
               "rt_hip_untile")
        return image, image8

    def render_image(self, seed, samples=None, max_depth=None, integrator="path"):
        """Whole image on this one GPU -> (image f32 [H,W,3], image8 u8 [H,W,3], stats dict), synchronised."""
        total = n_tiles(self.scene.width, self.scene.height)
        chunks = 1 if integrator != "path" else self.suggest_chunks(total, samples, max_depth)
        tiles, tiles8, stats = self.render_tiles(seed, 0, 1, total, samples=samples, max_depth=max_depth,
                                                 integrator=integrator, chunks=chunks)
        image, image8 = self.untile(tiles, tiles8, 0, 1, total)
        torch.cuda.synchronize(tiles.device)
        self.launch_status()   # a workgroup without its pool slot fails the frame here, not as NaN pixels
        st = stats.cpu().tolist()
        return image, image8, dict(rays=st[abi.STAT_RAYS], casts=st[abi.STAT_CASTS], tests=st[abi.STAT_TESTS],
                                   samples=st[abi.STAT_SAMPLES])


def render_image_host(scene, seed, n_devices=1, samples=None, max_depth=None, integrator="path"):
    """rt_hip_render_image(): the C hosts' entry point (host buffers, synchronous)."""
    import numpy as np
    shim = abi.load_shim()
    p = abi.RtHipParams()
    p.width, p.height = scene.width, scene.height
    p.samples = samples or scene.samples
    p.max_depth = scene.max_depth if max_depth is None else max_depth
    p.seed = seed
    p.integrator = abi.INTEGRATORS[integrator]
    img = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
    img8 = np.zeros((scene.height, scene.width, 3), dtype=np.uint8)
    stats = (C.c_uint64 * abi.NSTATS)()
    secs = C.c_double(0)
    meshes = scene.hip_meshes()
    _check(shim.rt_hip_render_image(scene.objects, scene.n_objects, meshes, scene.n_meshes, C.byref(scene.camera),
                                    C.byref(p), n_devices, img.ctypes.data, img8.ctypes.data, stats, C.byref(secs)),
           "rt_hip_render_image")
    return img, img8, dict(rays=stats[0], casts=stats[1], tests=stats[2], samples=stats[3]), secs.value
