#!/usr/bin/env python3
"""Where a wave's cycles go in the pooled kernels, phase by phase (PT_PHASE build: make variant NAME=phase DEFS="-DPT_PHASE").
usage: RT_HIP_SHIM_PATH=raytracer.c_amd/csrc/variants/librt_hip_phase.so python tools/phase.py [config] [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "raytracer.c_amd"))
import torch
from rt_amd import gpu as G, scene as S
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sc = S.build_scene(cfg, samples=spp)
gs = G.GpuScene(sc)
stats = torch.zeros(96, dtype=torch.int64, device="cuda")
for _ in range(3): # the first launch of a process pays cold TLBs and instruction fetch: profile a warm one
    gs.render_tiles(1666943821, 0, 1, G.n_tiles(sc.width, sc.height), stats=stats)
torch.cuda.synchronize()
stats.zero_()
gs.render_tiles(1666943821, 0, 1, G.n_tiles(sc.width, sc.height), stats=stats)
torch.cuda.synchronize()
st = stats.cpu().tolist()
ph = st[64:80]
names = ["camera samples of a swap (start_sample) + rest of the trip's head", "phase 1: filter (+ wall bounds, fp32 pre-tests)", "phase 2: exact tests", "hit record, roulette, material",
         "direction rounds", "radiance to the pixel sums", "idle lanes take waiting paths from the list", "the swap: batch counter, busy lanes to the list",
         "prologue: staging (scene -> LDS)", "prologue: pixel keys, camera, wall table", "prologue: first barrier", "prologue: tile_cull, its barrier, set-up",
         "epilogue: waiting for the workgroup's other waves", "epilogue: mean, tonemap, tile store",
         "parked-walk kernels: every path to the list before a walk", "parked-walk kernels: walking the parked rays"]
if "queued" in gs.kernel_name():
    names[11] = "parked-walk kernels: depth test, mesh probe, parking"
tot = sum(ph)
n_wg = G.n_tiles(sc.width, sc.height)
n_rep = (n_wg + 31) // 32 * 4 # waves that reported: one workgroup in 32
print(f"kernel {gs.kernel_name()}, config {cfg}, {spp} spp: {n_wg} workgroups, casts {st[1]}; a wave lives {tot / n_rep:.0f} cycles (elapsed, shared with the SIMD's other waves)")
for n, v in zip(names, ph):
    if v:
        print(f"  {n:50s} {100.0 * v / tot:5.1f} %")
