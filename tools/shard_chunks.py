#!/usr/bin/env python3
"""One rank's share of a frame -- every Nth tile -- rendered alone on one GPU, by sample chunks per tile: what
rt_hip_suggest_chunks_depth picks and the counts around it.  Scenes: bench.py's names (1..5, glass, glass_mesh).
Round 5: the M_REFRACTION forms of the pooled and parked-walk kernels take chunks too (round 4: one chunk, always).
usage: python tools/shard_chunks.py [N=8] [scene:spp ...]      default: glass:1024 glass_mesh:256 5:256 4:1024"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracer.c_amd")]
import torch
import bench
from rt_amd import gpu as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
specs = sys.argv[2:] or ["glass:1024", "glass_mesh:256", "5:256", "4:1024"]


def timed(gs, first, stride, count, chunks, reps=3):
    gs.render_tiles(bench.SEED, first, stride, count, chunks=chunks)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        gs.render_tiles(bench.SEED, first, stride, count, chunks=chunks)
        b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)


for spec in specs:
    name, spp = spec.split(":")
    sc = bench.make_scene(name, None, None, int(spp))
    gs = G.GpuScene(sc)
    total = G.n_tiles(sc.width, sc.height)
    count = (total + N - 1) // N
    sug = gs.suggest_chunks(count)
    full_ms = timed(gs, 0, 1, total, gs.suggest_chunks(total), reps=2)
    row = [(c, timed(gs, 0, N, count, c)) for c in sorted({1, 2, 4, 7, 8, 12, 16, sug}) if c <= sc.samples]
    print(f"{name:10s} {sc.width}x{sc.height} x {spp} spp, {gs.last_launch_kernel()}: whole frame {full_ms:.2f} ms -> ideal 1/{N} = {full_ms / N:.2f} ms; "
          f"rank 0's share ({count} tiles), ms by chunks: " + ", ".join(f"{c}: {ms:.2f}" for c, ms in row) + f"   (suggested: {sug})", flush=True)
    gs.close()
