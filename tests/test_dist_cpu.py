"""The N > 1 frame assembly on CPU: 2 (and 3) ranks over gloo.  Each rank fills the tile
buffer of ITS interleaved tile set (values from the CPU oracle, standing in for the kernel),
rt_amd.dist.gather_tiles() collects them on rank 0 -- the same function bench.py runs over
RCCL -- and the scattered image must equal the single-process image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import SEED

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, spp, out_path, mode="gather"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="1")
    for p in (os.path.join(ROOT, "raytracer.c_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import oracle_py
    from rt_amd import abi, dist as D, scene as S
    from util import tile_pixels, untile_numpy

    dist.init_process_group("gloo", rank=rank, world_size=world)
    D._MODE[0] = mode  # "all_gather": the fallback for a backend without gather-to-root
    sc = S.build_scene(2, w, h, spp)
    pt = oracle_py.PtOracle()
    first, stride, count = D.rank_tiles(w, h, rank, world)
    tiles, tiles8 = D.alloc_tile_buffers(w, h, world, torch.device("cpu"))
    assert tiles.shape[0] == D.padded_count(w, h, world) >= count
    tx = (w + 7) // 8
    for k in range(count):  # "render" my tiles into the compact tile-major layout
        t = first + k * stride
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        for pit in range(64):
            x, y = x0 + (pit & 7), y0 + (pit >> 3)
            if x < w and y < h:
                mean, rgb8, _ = pt.render_pixels(sc, SEED, pixels=np.array([y * w + x], dtype=np.uint32))
                tiles[k, pit] = torch.from_numpy(mean[0].astype(np.float32))
                tiles8[k, pit] = torch.from_numpy(rgb8[0])
    parts, parts8 = D.gather_tiles(tiles, tiles8, rank, world)
    if rank == 0:
        image = np.zeros((h, w, 3), dtype=np.float32)
        image8 = np.zeros((h, w, 3), dtype=np.uint8)
        for r, f, s_, c in D.segments(w, h, world):
            untile_numpy(parts[r].numpy(), w, h, f, s_, c, image)
            untile_numpy(parts8[r].numpy(), w, h, f, s_, c, image8)
        mean, rgb8, _ = pt.render_pixels(sc, SEED)
        ok = np.array_equal(image.reshape(-1, 3), mean.astype(np.float32)) and np.array_equal(image8.reshape(-1, 3), rgb8)
        open(out_path, "w").write("OK" if ok else "MISMATCH")
    else:
        assert parts is None and parts8 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 40, 24), (3, 37, 21), (8, 45, 27)])  # 8 ranks, ragged image: 6 x 4 = 24 tiles, 3 per rank
def test_gather_assembles_the_frame(tmp_path, world, w, h):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), w, h, 2, out), nprocs=world, join=True)
    assert open(out).read() == "OK"


def test_all_gather_fallback_assembles_the_same_frame(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), 40, 24, 2, out, "all_gather"), nprocs=2, join=True)
    assert open(out).read() == "OK"


def test_world_one_needs_no_collective():
    import torch
    from rt_amd import dist as D
    a, b = D.alloc_tile_buffers(16, 16, 1, torch.device("cpu"))
    parts, parts8 = D.gather_tiles(a, b, 0, 1)
    assert parts[0] is a and parts8[0] is b


def test_force_collective_with_one_rank(tmp_path):
    """world 1 normally short-circuits; force_collective sends the one-rank group through the backend's gather
    (what tests/rccl_child.py does over RCCL on the GPU box)"""
    import torch
    import torch.distributed as dist
    from rt_amd import dist as D
    dist.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{_free_port()}")
    try:
        tiles = torch.arange(3 * 64 * 3, dtype=torch.float32).reshape(3, 64, 3)
        tiles8 = (torch.arange(3 * 64 * 3) % 251).to(torch.uint8).reshape(3, 64, 3)
        same, same8 = D.gather_tiles(tiles, tiles8, 0, 1)
        assert same[0] is tiles and same8[0] is tiles8
        parts, parts8 = D.gather_tiles(tiles, tiles8, 0, 1, force_collective=True)
        assert parts[0] is not tiles and torch.equal(parts[0], tiles) and torch.equal(parts8[0], tiles8)
    finally:
        dist.destroy_process_group()


def test_bench_started_without_a_launcher_and_without_gpus():
    """`python bench.py --gpus 8` as the driver calls it, on a box without GPUs: no traceback, one JSON line that
    says why, exit code 2 (with GPUs it starts its own ranks: tests/test_gpu_multi.py)"""
    import json
    import subprocess
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RT_BENCH_REHEARSE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 2, p.stderr[-1000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 8 and "error" in json.loads(lines[0])
