#!/usr/bin/env python3
"""Child process of tests/test_gpu_multi.py: RCCL with ONE rank on this box's one GPU.

mode `torch`: torch.distributed's "nccl" backend (= RCCL on ROCm), world size 1: init_process_group with device_id,
              an all_reduce, and rt_amd.dist.gather_tiles() forced through dist.gather (force_collective) -- the frame
              assembled from the gathered buffers must equal the plain one bit for bit.
mode `shim` : RT_HIP_FORCE_COMM=1 makes rt_hip_render_image(n_devices = 1) create its cached communicator with
              ncclCommInitAll(.., 1, ..), send device 0's tile buffers to itself through the grouped ncclSend / ncclRecv
              block of the N > 1 path, and destroy the communicator in rt_hip_release_cache().
mode `shim_map`: the same switch under rt_hip_set_device_map({0 x 8}): 1, 2, 3 and 8 LOGICAL devices on the one GPU, one
              communicator (a physical device is one rank), every segment sent to self through RCCL.
Prints one JSON line.  A separate process so that a bootstrap problem (no NIC, library mismatch) or a hang is the
child's, under the parent's time-out."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "raytracer.c_amd"))
SEED = 1666943821


def torch_mode():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29683"), RANK="0", WORLD_SIZE="1")
    import torch
    import torch.distributed as dist
    from rt_amd import dist as D, gpu as G, scene as S
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    one = torch.ones(4, dtype=torch.int32, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize(dev)
    sc = S.build_scene(2, 200, 120, 4)
    gs = G.GpuScene(sc, device=0)
    W, H = sc.width, sc.height
    first, stride, count = D.rank_tiles(W, H, 0, 1)
    tiles, tiles8 = D.alloc_tile_buffers(W, H, 1, dev)
    gs.render_tiles(SEED, first, stride, count, tiles, tiles8)
    parts, parts8 = D.gather_tiles(tiles, tiles8, 0, 1, force_collective=True)
    image, image8 = gs.untile(parts[0], parts8[0], first, stride, count)
    ref, ref8, _ = gs.render_image(SEED)
    torch.cuda.synchronize(dev)
    out = {"mode": "torch", "backend": dist.get_backend(), "all_reduce": one.cpu().tolist(),
           "gathered_is_a_copy": parts[0].data_ptr() != tiles.data_ptr(),
           "frame_equal": bool(torch.equal(image, ref) and torch.equal(image8, ref8)), "gather_mode": D._MODE[0]}
    dist.barrier()
    dist.destroy_process_group()
    gs.close()
    print(json.dumps(out), flush=True)


def shim_mode():
    import numpy as np
    import torch  # noqa: F401  (one HIP runtime: before the shim)
    from rt_amd import abi, gpu as G, scene as S
    shim = abi.load_shim()
    sc = S.build_scene(2, 200, 120, 4)
    os.environ.pop("RT_HIP_FORCE_COMM", None)
    plain = G.render_image_host(sc, SEED, n_devices=1)
    b0 = shim.rt_hip_cache_builds()
    os.environ["RT_HIP_FORCE_COMM"] = "1"
    a = G.render_image_host(sc, SEED, n_devices=1)        # context rebuilt with a communicator
    b = G.render_image_host(sc, SEED, n_devices=1)        # ... and reused, communicator included
    builds = shim.rt_hip_cache_builds() - b0
    shim.rt_hip_release_cache()                           # ncclCommDestroy
    c = G.render_image_host(sc, SEED, n_devices=1)        # a second communicator in the same process
    shim.rt_hip_release_cache()
    del os.environ["RT_HIP_FORCE_COMM"]
    eq = all(np.array_equal(x[0], plain[0]) and np.array_equal(x[1], plain[1]) and x[2] == plain[2] for x in (a, b, c))
    print(json.dumps({"mode": "shim", "frame_equal": bool(eq), "context_builds_with_comm": int(builds)}), flush=True)


def shim_map_mode():
    import ctypes as C
    import numpy as np
    import torch  # noqa: F401
    from rt_amd import abi, gpu as G, scene as S
    shim = abi.load_shim()
    sc = S.build_scene(3, 204, 116, 4)
    os.environ.pop("RT_HIP_FORCE_COMM", None)
    plain = G.render_image_host(sc, SEED, n_devices=1)
    shim.rt_hip_release_cache()
    os.environ["RT_HIP_FORCE_COMM"] = "1"
    m = (C.c_int * 8)(*([0] * 8))
    assert shim.rt_hip_set_device_map(m, 8) == 0
    b0 = shim.rt_hip_cache_builds()
    devices, eq = [1, 2, 3, 8], True
    for g in devices:
        x = G.render_image_host(sc, SEED, n_devices=g)
        eq = eq and np.array_equal(x[0], plain[0]) and np.array_equal(x[1], plain[1]) and x[2] == plain[2]
    builds = shim.rt_hip_cache_builds() - b0
    shim.rt_hip_release_cache()
    shim.rt_hip_set_device_map(None, 0)
    del os.environ["RT_HIP_FORCE_COMM"]
    print(json.dumps({"mode": "shim_map", "frame_equal": bool(eq), "devices": devices, "context_builds": int(builds)}), flush=True)


if __name__ == "__main__":
    {"torch": torch_mode, "shim": shim_mode, "shim_map": shim_map_mode}[sys.argv[1]]()
