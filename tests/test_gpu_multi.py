"""The N > 1 machinery on the one GPU a test box has: RCCL executed with one rank (torch.distributed "nccl" and the
shim's own communicators), and bench.py started the way the driver starts it (`python bench.py --gpus N`, no
launcher).  The assembly logic itself is covered with 2, 3 and 8 gloo ranks in tests/test_dist_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(args, env=None, timeout=300):
    p = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_one_rank_through_torch_distributed():
    """init_process_group("nccl", world_size=1), an all_reduce, and the frame's gather forced through dist.gather"""
    d = _child([os.path.join(ROOT, "tests", "rccl_child.py"), "torch"])
    assert d["backend"] == "nccl" and d["all_reduce"] == [1, 1, 1, 1]
    assert d["gathered_is_a_copy"], "the gather must have gone through the backend, not returned the input"
    assert d["frame_equal"]


def test_rccl_one_device_through_the_shim():
    """RT_HIP_FORCE_COMM=1: ncclCommInitAll / grouped ncclSend + ncclRecv / ncclCommDestroy of rt_hip_render_image"""
    d = _child([os.path.join(ROOT, "tests", "rccl_child.py"), "shim"])
    assert d["frame_equal"]
    assert d["context_builds_with_comm"] == 1, "forcing the communicator rebuilds the context once; the next frame reuses it"


def test_bench_gpus_2_started_directly_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (the driver's command shape): the parent starts
    torch.distributed.run as a child and relays rank 0's one JSON line.  Two ranks share this box's GPU, so the
    rehearsal switch is on (gloo, not a measurement)."""
    d = _child([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "2", "--spp", "4", "--steps", "1", "--warmup", "1",
                "--cpu-tiles", "0", "--no-configs"], env={"RT_BENCH_REHEARSE": "1"})
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 1
    assert "REHEARSAL" in d["config"]["parallelism"] and d["value"] > 0
    assert len(d["rank_kernel_ms"]["per_rank"]) == 2


def test_bench_four_ranks_rehearsal_assembles_the_one_gpu_frame():
    """four ranks sharing this box's GPU (gloo): the interleaved partition at a rank count that leaves the ranks unequal
    tile counts on config 1's 32 x 32 tiles + ragged edge, gathered, scattered, and compared on rank 0 with a one-GPU render of
    a strided sample -- `assembly_check` at more than two ranks.  (Not eight: a GPU box admits six processes on its card, this
    test process is one of them; the eight-rank assembly runs on the CPU over gloo in tests/test_dist_cpu.py, and eight
    LOGICAL devices run through the C host's path in tests/test_gpu_devmap.py.)"""
    d = _child([os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "1", "--width", "250", "--height", "250", "--spp", "4",
                "--steps", "1", "--warmup", "1", "--cpu-tiles", "0", "--no-configs"], env={"RT_BENCH_REHEARSE": "1"}, timeout=400)
    assert d["n_gpus"] == 4 and d["ranks_seen"] == 4 and "REHEARSAL" in d["config"]["parallelism"]
    assert d["assembly_check"]["bit_identical_to_one_gpu"] is True and d["assembly_check"]["tiles"] == 1024
    assert len(d["rank_kernel_ms"]["per_rank"]) == 4


def test_bench_gpus_n_without_enough_gpus_says_so():
    """more ranks than GPUs and no rehearsal switch: one JSON line with an error, non-zero exit, nothing launched"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k != "RT_BENCH_REHEARSE"})
    assert p.returncode == 2
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 64 and "GPU(s) visible" in d["error"]
