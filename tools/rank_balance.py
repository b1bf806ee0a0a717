#!/usr/bin/env python3
"""Load balance of the N-way interleaved tile partition: every rank's share of a frame rendered alone on one GPU.
(With tools/experiments/r05_tile_skew.patch applied -- render_tiles(skew=) exists -- also with the skewed, diagonal partition.)
usage: python tools/rank_balance.py [scene:spp:N ...]      default: 5:1024:8 4:1024:8 5:1024:4"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracer.c_amd")]
import torch, bench
from rt_amd import dist as D, gpu as G

for spec in (sys.argv[1:] or ["5:1024:8", "4:1024:8", "5:1024:4"]):
    name, spp, N = spec.split(":")
    spp, N = int(spp), int(N)
    sc = bench.make_scene(name, None, None, spp)
    gs = G.GpuScene(sc)
    total = G.n_tiles(sc.width, sc.height)
    have_skew = hasattr(D, "partition_skew")
    for skew in (sorted({0, D.partition_skew(sc.width, N)}) if have_skew else [0]):
        ms = []
        for r in range(N):
            count = (total - r + N - 1) // N
            ch = gs.suggest_chunks(count)
            kw = dict(skew=skew) if have_skew else {}
            gs.render_tiles(bench.SEED, r, N, count, chunks=ch, **kw)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            gs.render_tiles(bench.SEED, r, N, count, chunks=ch, **kw)
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        print(f"{name} x {spp} spp, N = {N}, {ch} chunks, tile_skew {skew}: per-rank ms " + " ".join(f"{m:.2f}" for m in ms) +
              f"   max {max(ms):.2f}  mean {sum(ms) / N:.2f}  max/mean {max(ms) / (sum(ms) / N):.4f}", flush=True)
    gs.close()
