#!/usr/bin/env python3
"""bench.py -- the headline benchmark: ray-bounces/s (+ Mpixel-samples/s) of the path-tracing
hot path on BASELINE.json configs[3]: 1920x1080, 1024 spp, the reference's 38-sphere
Cornell-box-style room, depth 16.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full frame.  With N ranks the frame's 8x8 tiles are interleaved over the
ranks (rank r renders tiles r, r+N, ...: no data-path collective), then the compact tile
buffers are gathered to rank 0 over RCCL and scattered to the row-major image there.  Total
work is fixed as N grows ("scaling": "strong").  Inputs (scene, camera) are resident in HBM
before the timed region; the timed region is K x (render [+ gather] + untile), bracketed by
barrier + synchronize, MAX over ranks.

The JSON line also carries
  roofline      fp64-VALU roofline of the render kernel (the binding roof of this branchy
                fp64 path) from HIP-event kernel times measured here;
  roofline_hbm  the same launch against the HBM roof (reported because the north star asks
                for it; ~1e-5 by construction);
  cpu_baseline  the reference's own compiled trace_path()/intersect() (oracle/_ref) timed on
                this host's cores over a bounded sample of the same frame (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "raytracer.c_amd"))

SEED = 1666943821            # reference main.c:182
PEAK_FP64_TFLOPS = 39.3      # 256 CU x 4 SIMD x 16 fp64 lanes x 2.4 GHz, one op per lane-slot (no FMA credit)
PEAK_HBM_GBS = 8000.0


def measured_traffic(config, width, height, spp, world):
    """HBM bytes per launch from the committed PMC passes (profiles/), when this run is the
    profiled configuration; None otherwise (PMC counters cannot be collected from inside)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic_c4.json")))
        if (rec["config"], rec["width"], rec["height"], rec["spp"], rec["n_gpus"]) == (config, width, height, spp, world):
            return rec["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def flops_per_ray(n_spheres, n_triangles):
    """SURVEY.md section 8d counting convention (add/sub/mul/div/sqrt = 1, no FMA credit)."""
    return 17.0 * n_spheres + 40.0 * n_triangles + 120.0


# ---- CPU baseline: the compiled reference on a bounded sample -----------------------------

def _cpu_worker(args):
    config, width, height, spp, depth, pixels, kind = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from rt_amd import scene as S
    sc = S.build_scene(config, width, height, spp)
    if kind == "reference":
        _, _, st = oracle_py.RefOracle(depth).render_pixels(sc, SEED, pixels=pixels, want_rgb8=False)
        casts = st["tests"] // max(sc.n_primitives, 1)
    else:
        _, _, st = oracle_py.PtOracle().render_pixels(sc, SEED, pixels=pixels, want_rgb8=False)
        casts = st["casts"]
    return casts, st["rays"]


def cpu_baseline(config, width, height, spp, depth, n_meshes, budget_tiles):
    """Times the CPU checker on `budget_tiles` 8x8 tiles spread evenly over the frame at full
    spp, one single-threaded process per core (processes, not threads: the reference's global
    counters make its threads slower)."""
    import multiprocessing as mp
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    kind = "reference" if (n_meshes == 0 and oracle_py.ref_available(depth)) else "port"
    tx, ty = (width + 7) // 8, (height + 7) // 8
    total = tx * ty
    tiles = np.unique(np.linspace(0, total - 1, num=min(budget_tiles, total)).astype(np.int64))
    px = []
    for t in tiles:
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        for r in range(8):
            for c in range(8):
                if x0 + c < width and y0 + r < height:
                    px.append((y0 + r) * width + x0 + c)
    px = np.array(px, dtype=np.uint32)
    cores = min(os.cpu_count() or 1, 64)
    chunks = [px[i::cores] for i in range(cores)]  # interleaved: even load
    jobs = [(config, width, height, spp, depth, ch, kind) for ch in chunks if len(ch)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(len(jobs)) as pool:
        pool.map(_cpu_worker, [(config, width, height, 1, depth, ch[:1], kind) for ch in chunks if len(ch)])  # warm
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, jobs)
        dt = time.perf_counter() - t0
    casts = sum(r[0] for r in res)
    return {"value": casts / dt, "unit": "ray-bounces/s", "cores": len(jobs), "kind": kind,
            "sample": f"{len(tiles)} of {total} 8x8 tiles ({len(px)} pixels) at full {spp} spp, "
                      f"{len(jobs)} single-thread processes, {dt:.1f} s wall",
            "mpixel_samples_per_s": len(px) * spp / dt * 1e-6}


def cpu_as_shipped():
    """BASELINE.md CPU-A: the reference's render() exactly as shipped (libc rand(), its own
    compile-time MAX_DEPTH 5, one OpenMP thread = its deterministic and fastest mode)
    on its own scene at 320x180x16 -- context for readers of the reference, not the baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from rt_amd import scene as S
    if not oracle_py.ref_available(5):
        return None
    sc = S.build_scene(4, 320, 180, 16, max_depth=5)
    small = S.build_scene(4, 160, 90, 16, max_depth=5)
    ref = oracle_py.RefOracle(5)
    threads = min(os.cpu_count() or 1, 64)
    devnull = os.open(os.devnull, os.O_WRONLY)       # render() prints a progress bar
    saved = os.dup(1)
    os.dup2(devnull, 1)
    try:
        t0 = time.perf_counter()
        _, st = ref.render_as_shipped(sc, SEED, threads=1)
        dt = time.perf_counter() - t0
        # CPU-B: the same render() with its OpenMP team on every core.  rand() is one locked global
        # stream and the two counters one contended cache line, so this is the slower way
        # to run it -- shown because it is what `make && ./raytracer` does on a many-core host.
        t0 = time.perf_counter()
        _, st_all = ref.render_as_shipped(small, SEED, threads=threads)
        dt_all = time.perf_counter() - t0
        ref.lib.ref_set_threads(1)
    finally:
        import ctypes
        ctypes.CDLL(None).fflush(None)               # its printf buffer must drain into /dev/null, not after our JSON
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)
    return {"value": st["tests"] / sc.n_objects / dt, "unit": "ray-bounces/s", "cores": 1, "kind": "reference",
            "sample": f"render() as shipped, 320x180, 16 spp, MAX_DEPTH 5, libc rand(), 1 thread, {dt:.1f} s",
            "all_cores": {"value": st_all["tests"] / small.n_objects / dt_all, "unit": "ray-bounces/s", "cores": threads,
                          "sample": f"render() as shipped, 160x90, 16 spp, MAX_DEPTH 5, libc rand(), {threads} OpenMP "
                                    f"threads, {dt_all:.1f} s (racy counters: the count is approximate)"}}


# ---- main ------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=4, help="BASELINE.json configs index + 1 (default 4: the headline)")
    ap.add_argument("--spp", type=int, default=0, help="override samples/pixel (0 = the configuration's own)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--chunks", type=int, default=0, help="sample chunks per tile (0 = rt_hip_suggest_chunks)")
    ap.add_argument("--shard", type=str, default="", help="dev: render only rank R of a W-way partition, 'R/W', on this one GPU")
    ap.add_argument("--cpu-tiles", type=int, default=384, help="8x8 tiles of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration array")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rt_amd import abi, dist as D, gpu as G, scene as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available() or abi.load_shim().rt_hip_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RT_BENCH_REHEARSE=1: every rank on GPU 0, gloo instead of RCCL, the gather staged through the
    # host -- the N > 1 control flow on a one-GPU box (RCCL refuses two ranks on one device).  The
    # number it prints is not a measurement; the JSON says so.
    rehearse = world > 1 and os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    sc = S.build_scene(args.config, args.width or None, args.height or None, args.spp or None)
    W, H, spp, depth = sc.width, sc.height, sc.samples, sc.max_depth
    gs = G.GpuScene(sc, device=local_rank)            # scene resident in HBM from here on
    first, stride, count = D.rank_tiles(W, H, rank, world)
    if args.shard:  # development aid: what one rank of a larger job would do
        r_, w_ = [int(v) for v in args.shard.split("/")]
        first, stride, count = D.rank_tiles(W, H, r_, w_)
    tiles, tiles8 = D.alloc_tile_buffers(W, H, world, dev)   # padded: ranks differ by at most one tile
    stats = torch.zeros(abi.NSTATS, dtype=torch.int64, device=dev)
    # few tiles per GPU (large N): split every tile's samples over several workgroups so the
    # last, partly filled round of the launch stays a small fraction of it (same image, bit for bit)
    chunks = args.chunks or gs.suggest_chunks(count)
    workspace = (torch.empty(abi.load_shim().rt_hip_chunk_workspace_bytes(max(count, 1)), dtype=torch.uint8, device=dev)
                 if chunks > 1 else None)
    image = torch.zeros((H, W, 3), dtype=torch.float32, device=dev) if rank == 0 else None
    image8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
    gathered = gathered8 = None
    if world > 1 and rank == 0:
        gathered = [torch.empty_like(tiles) for _ in range(world)]
        gathered8 = [torch.empty_like(tiles8) for _ in range(world)]
    kernel_events = []

    def step(timed):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()                                    # same stream the shim launches on
        gs.render_tiles(SEED, first, stride, count, tiles, tiles8, stats, chunks=chunks, workspace=workspace)
        e1.record()
        if timed:
            kernel_events.append((e0, e1))
        # RCCL over xGMI when world > 1: the one exchange of the path
        parts, parts8 = D.gather_tiles(tiles, tiles8, rank, world, gathered, gathered8, via_cpu=rehearse)
        if rank == 0:
            for r, f, s_, c in D.segments(W, H, world):
                gs.untile(parts[r], parts8[r], f, s_, c, image, image8)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    fence()
    stats.zero_()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    kern_ms = torch.tensor([sum(a.elapsed_time(b) for a, b in kernel_events) / max(len(kernel_events), 1)],
                           dtype=torch.float64, device=dev)
    tot = stats.clone()
    if rehearse:
        t, kern_ms, tot = t.cpu(), kern_ms.cpu(), tot.cpu()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(kern_ms, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    kern_s = float(kern_ms.item()) * 1e-3
    rays, casts, tests, samples = [int(v) for v in tot.cpu().tolist()]

    if rank == 0:
        steps = max(args.steps, 1)
        casts_per_step = casts / steps
        fr = flops_per_ray(sc.n_objects, sc.n_triangles)
        # dominant kernel = pt_render_tiles; its work per launch on the slowest rank ~ 1/world of the frame
        launch_casts = casts_per_step / world
        achieved_tflops = launch_casts * fr / kern_s * 1e-12 if kern_s > 0 else 0.0
        alg_bytes = (12 + 3) * W * H / world + 88 * sc.n_objects + 72 * sc.n_triangles
        out = {
            "metric": "ray-bounces/sec", "value": casts / elapsed, "unit": "ray-bounces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: {W}x{H}, {spp} spp, "
                                   f"{sc.n_objects} spheres + {sc.n_triangles} triangles, depth {depth}",
                       "width": W, "height": H, "spp": spp, "max_depth": depth, "seed": SEED,
                       "parallelism": (f"tiles interleaved over {world} GPU(s), RCCL gather to rank 0" if not rehearse else
                                       f"REHEARSAL: {world} ranks sharing one GPU, gloo gather through the host (not a measurement)"),
                       "sample_chunks_per_tile": chunks},
            "mpixel_samples_per_s": samples / elapsed * 1e-6,
            "rays_per_sample": rays / max(samples, 1),
            "ray_count_per_step": rays / steps, "ray_bounces_per_step": casts_per_step,
            "intersection_tests_per_s": tests / elapsed,
            "roofline": {"bound": "valu_fp64", "kernel": gs.kernel_name(), "achieved": achieved_tflops,
                         "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tflops / PEAK_FP64_TFLOPS,
                         "traffic": measured_traffic(args.config, W, H, spp, world), "flops_per_ray_bounce": fr,
                         "kernel_ms": kern_s * 1e3,
                         "note": "branchy fp64 scalar-per-lane math: neither HBM nor MFMA binds it "
                                 "(BASELINE.md section 4); algorithmic flops, no FMA credit"},
            "roofline_hbm": {"bound": "hbm", "kernel": "pt_render_tiles",
                             "achieved": alg_bytes / kern_s * 1e-9 if kern_s > 0 else 0.0, "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": (alg_bytes / kern_s * 1e-9 / PEAK_HBM_GBS) if kern_s > 0 else 0.0,
                             "traffic": measured_traffic(args.config, W, H, spp, world),
                             "traffic_source": "profiles/traffic_c4.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                             "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and args.cpu_tiles > 0:
            try:
                if args.config == 4:
                    out["cpu_as_shipped"] = cpu_as_shipped()
                out["cpu_baseline"] = cpu_baseline(args.config, W, H, spp, depth, sc.n_meshes, args.cpu_tiles)
            except Exception as exc:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "ray-bounces/s", "cores": 0, "kind": "reference",
                                       "sample": f"failed: {exc}"}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    gs.close()


if __name__ == "__main__":
    main()
