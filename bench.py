#!/usr/bin/env python3
"""bench.py -- the headline benchmark: ray-bounces/s (+ Mpixel-samples/s) of the path-tracing
hot path on BASELINE.json configs[3]: 1920x1080, 1024 spp, the reference's 38-sphere
Cornell-box-style room, depth 16.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nproc-per-node N ...
     bench.py --gpus N ... -- or plain `python bench.py --gpus N`, which starts its own ranks
     in a child torch.distributed.run, see launch_ranks)

One "step" = one full frame.  With N ranks the frame's 8x8 tiles are interleaved over the
ranks (rank r renders tiles r, r+N, ...: no data-path collective), then the compact tile
buffers are gathered to rank 0 over RCCL and scattered to the row-major image there.  Total
work is fixed as N grows ("scaling": "strong").  Inputs (scene, camera) are resident in HBM
before the timed region; the timed region is K x (render [+ gather] + untile), bracketed by
barrier + synchronize, MAX over ranks.

The JSON line also carries
  roofline      fp64-VALU roofline of the render kernel (the binding roof of this branchy
                fp64 path) from HIP-event kernel times measured here.
                `achieved` counts SURVEY section 8d's ALGORITHMIC flops (what the reference's
                scan would execute), not executed fp64 instructions -- the kernel's packed-fp32
                filter skips most exact tests -- so the hardware-true statement rides along:
                VALU busy / lane utilisation / VALU instructions per 64 ray-bounces from the
                committed PMC pass of this configuration (profiles/pmc_c<N>.json);
  roofline_hbm  the same launch against the HBM roof (reported because the north star asks
                for it; ~1e-5 by construction);
                `frac_executed` prices the arithmetic the kernels' own algorithm executes (lane-level
                event counts of a committed PT_DIAG pass x flops per event, fp64 and fp32 each at its pipe's
                rate: EXEC_FLOPS), `frac_valu_lanes` = VALU busy x lanes active (the hardware-true share);
                PMC / PT_DIAG / ISA figures come from committed files under profiles/ and are reported only
                while the sha256 of the kernel sources they were measured on is today's and (PMC) the kernel
                time agrees within 3 % -- otherwise null with pmc_stale / diag_stale / isa_stale;
  cpu_baseline  the reference's own compiled trace_path()/intersect() (oracle/_ref) timed on
                this host's cores over a bounded sample of the same frame (rank 0, N = 1 only);
  parity        (N = 1) the pixels that CPU leg rendered -- 2,048 tiles at the full 1024 spp -- against the
                frame the timed steps produced: per-channel RMS, max |diff|, tonemapped-byte difference, and
                rays / ray-bounces of that subset equal to a GPU re-render of exactly those tiles; the run
                exits non-zero when it fails (north star: "within 1e-4 per-channel RMS ... in the same run");
  configs       (N = 1) the other BASELINE configurations on the same GPU, all at their own sizes --
                ms per frame, ray-bounces/s, kernel, fractions, and a parity block each;
  integrators   (N = 1) trace_path's M_REFRACTION branch (a glass scene; config 5's mesh turned to glass, 4K) and cast_ray, 1920x1080, with
                kernel, registers / scratch, and a parity block each;
  assembly_check (N > 1) the frame gathered from the N ranks against rank 0's own render of 2,048 of its tiles, bit for bit;
  phase_ms, ranks_seen, rank_kernel_ms   (N > 1) where a frame's time goes, how many ranks RCCL really connected;
  host_path     the single-process C path -- rt_hip_render_image, what render() of the raytracer.h boundary calls -- timed in a
                child process: N > 1 over the N devices (grouped RCCL send / recv inside the shim), N = 1 the headline's
                whole-call time (scene compare, launches, the frame over PCIe; never `value`) with its phase split;
  host_path_logical8  (N = 1) the same frame on 8 LOGICAL devices mapped onto this GPU (rt_hip_set_device_map): the C host's
                multi-device code path executed on the one GPU there is, its frame compared bit for bit with the one-device
                frame (not a scaling measurement);
  cli_host      (N = 1) the C command-line host as a process: wall time from start to exit for this workload, by phase
                (HIP runtime start, context, render, copy-out, PNG);
  configs[config 5].parity.wide   65,536 pixels of the 3840x2160 frame at 4 spp against the compiled reference (tiles on the
                mesh's outline, inside it, and elsewhere), next to the two tiles at the full 4096 spp.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "raytracer.c_amd"))

SEED = 1666943821            # reference main.c:182
PEAK_FP64_TFLOPS = 39.3      # 256 CU x 4 SIMD x 16 fp64 lanes x 2.4 GHz, one op per lane-slot (no FMA credit)
PEAK_FP32_TFLOPS = 157.3     # vector fp32 with fused multiply-adds (MI355X_MICROARCH.md), the rate of the packed-fp32 filter / slab tests
PEAK_HBM_GBS = 8000.0
FLOPS_NOTE = ("algorithmic model flops (SURVEY 8d: 17 per sphere test + 40 per triangle test + 120 per ray-bounce, no FMA "
              "credit), NOT executed fp64 instructions: the packed-fp32 filter / the hierarchy skip most exact tests, so "
              "`frac` says how fast the reference's work gets done, not how full the fp64 pipe is; see valu_busy / "
              "lane_utilisation for the hardware-true picture")


# the device code's translation unit, every header it is made of (pt_math.h ... pt_body_static.h: one TU, split by topic), the
# shim that picks and launches the kernels, and the two public headers both see
KERNEL_SOURCES = ("raytracer.c_amd/csrc/pt_kernel.hip", "raytracer.c_amd/csrc/pt_device.h", "raytracer.c_amd/csrc/pt_math.h",
                  "raytracer.c_amd/csrc/pt_intersect.h", "raytracer.c_amd/csrc/pt_filter.h", "raytracer.c_amd/csrc/pt_scene_ctx.h",
                  "raytracer.c_amd/csrc/pt_trace.h", "raytracer.c_amd/csrc/pt_body_pooled.h", "raytracer.c_amd/csrc/pt_body_queued.h",
                  "raytracer.c_amd/csrc/pt_body_static.h", "raytracer.c_amd/csrc/rt_hip_shim.hip", "raytracer.c_amd/csrc/bvh_build.h", "include/rt_rng.h", "include/rt_hip.h")


def kernel_source_sha256():
    """sha256 over the device code's sources: what ties a committed PMC / PT_DIAG summary under profiles/ to the
    kernels it was measured on (tools/summarize_pmc_cfg.py and tools/diag.py write it, bench.py compares)."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def pmc_is_stale(pmc, kernel_ms, spp, source_sha=None):
    """-> None if the committed PMC record still describes the kernel that was just timed, else the reason.
    Two checks: the record's source hash must be today's (a kernel edit without a re-profile), and the kernel time
    of the PMC pass must agree with the time measured now -- within 3 % (+ 20 us) at the same spp, within 8 % per sample
    when the pass ran at another spp (frame time is not exactly linear in spp)."""
    if not pmc:
        return None
    have = pmc.get("source_sha256")
    want = source_sha if source_sha is not None else kernel_source_sha256()
    if have != want:
        return f"source hash of the PMC pass ({str(have)[:12]}) is not that of the kernel sources now ({want[:12]})"
    ms = pmc.get("kernel_ms")
    if ms is None or not kernel_ms:
        return "the PMC record carries no kernel time"
    if pmc["spp"] == spp:
        tol, ref = 0.03, ms
    else:
        tol, ref = 0.08, ms * spp / pmc["spp"]
    # (+ 20 microseconds: a sub-millisecond kernel's time moves by that much with the clock's ramp from one process to the
    # next -- config 2's 0.88 ms measured 0.872 ... 0.895 within one session)
    if abs(kernel_ms - ref) > tol * ref + 0.02:
        return f"kernel time now {kernel_ms:.3f} ms vs {ref:.3f} ms in the PMC pass (more than {tol:.0%} + 0.02 ms apart)"
    return None


# ---- executed-work model -------------------------------------------------------------
# `frac` prices the reference's ALGORITHM (every primitive tested exactly for every ray-bounce).
# `frac_executed` prices the arithmetic the kernels' own algorithm executes, from lane-level event counts of a
# committed PT_DIAG pass of the same configuration (profiles/diag_c<N>.json, tools/diag.py --json): flops by
# the same counting convention (add / sub / mul / div / sqrt = 1 each, so one FMA = 2; compares, selects and
# integer work = 0), each class at the rate of the pipe it runs on:
#   fp64 (contraction off: one flop per lane-slot)                39.3 T/s   (256 CU x 4 SIMD x 16 x 2.4 GHz)
#   fp32 (filter, slab tests, pre-tests: fused multiply-adds)    157.3 T/s   (MI355X_MICROARCH.md: vector fp32 peak)
# Flops per event, from the reference's arithmetic (file:line of raytracer.c) or from the kernel's fp32 form:
EXEC_FLOPS = {
    # fp64
    "camera_sample": (35.0, "f64"),     # :203-206 + get_camera_ray :375-384: 2 quotients, 15 add/mul, dot, sqrt, div, 3 mul
    "exact_sphere": (17.0, "f64"),      # intersect_sphere :82-117, SURVEY's nominal (8 / 16 / 20 at its three exits)
    "exact_triangle": (40.0, "f64"),    # intersect_triangle :132-150, SURVEY's nominal (20 / 30 / 45 / 51)
    "hit": (29.0, "f64"),               # point_at 6 + normal 13 (:408-409) + roulette scale 4 + radiance term 6
    "rejection_round": (11.0, "f64"),   # :239 three r * 2 - 1 (6) + the squared length (5); the sqrt leaves the loop
    "direction": (27.0, "f64"),         # :242-253, :549-553: normalise 13, dot 5, throughput albedo * cos 3 + 6
    # fp32
    "filter_sphere": (15.0, "f32"),     # conservative tca (3 mul + 3 add), |L|^2 - r2 (3 mul + 4 add), q = tca |tca| - ll (2)
    "node_visit": (36.0, "f32"),        # two child boxes: 12 plane distances (sub, mul) = 24, widening 4 x (mul, add) = 8, + 4
    "leaf_pretest": (51.0, "f32"),      # Moeller-Trumbore in fp32: 2 cross (18), 4 dot (20), s = o - v0 (3), thresholds (10)
    "probe": (13.0, "f32"),             # the triangles' bounding sphere: L (3), tca (5), |L|^2 (5); boxes are the walk's
}


def committed_diag(config):
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", f"diag_c{config}.json")))
        rec["file"] = f"profiles/diag_c{config}.json"
        return rec
    except Exception:
        return None


def executed_work(diag, casts, kernel_s, n_spheres, n_triangles, source_sha=None):
    """-> dict(frac_executed, flops_f64_per_ray_bounce, flops_f32_per_ray_bounce, frac_hierarchy, source...) from a
    committed PT_DIAG record, or nulls with the reason."""
    none = {"frac_executed": None, "executed_source": None}
    if not diag:
        none["executed_note"] = "no committed PT_DIAG record of this configuration"
        return none
    want = source_sha if source_sha is not None else kernel_source_sha256()
    if diag.get("source_sha256") != want:
        none["diag_stale"] = True
        none["executed_note"] = (f"stale: {diag['file']} was counted on sources {str(diag.get('source_sha256'))[:12]}, "
                                 f"the kernel sources now hash to {want[:12]}")
        return none
    pc = diag["per_ray_bounce"]
    f64 = f32 = 0.0
    terms = {}
    for key, (flops, pipe) in EXEC_FLOPS.items():
        n = float(pc.get(key, 0.0))
        terms[key] = n
        if pipe == "f64":
            f64 += n * flops
        else:
            f32 += n * flops
    rate = casts / kernel_s if kernel_s > 0 else 0.0
    out = {"frac_executed": rate * (f64 / (PEAK_FP64_TFLOPS * 1e12) + f32 / (PEAK_FP32_TFLOPS * 1e12)),
           "executed_flops_per_ray_bounce": {"f64": f64, "f32": f32},
           "executed_events_per_ray_bounce": terms,
           "executed_source": f"{diag['file']} (PT_DIAG lane-level event counts at {diag['width']}x{diag['height']}, "
                              f"{diag['spp']} spp, kernel {diag['kernel']}); flops per event and pipe rates: bench.py EXEC_FLOPS"}
    if n_triangles > 256:
        # the algorithmic model with the hierarchy in place of the O(N) triangle scan (raytracer.c:401-435): spheres and
        # shading as SURVEY 8d counts them, the triangles by what the walk executes
        alg64 = 17.0 * n_spheres + 120.0 + terms["exact_triangle"] * 40.0
        alg32 = terms["node_visit"] * EXEC_FLOPS["node_visit"][0] + terms["leaf_pretest"] * EXEC_FLOPS["leaf_pretest"][0] + \
            terms["probe"] * EXEC_FLOPS["probe"][0]
        out["frac_hierarchy_model"] = rate * (alg64 / (PEAK_FP64_TFLOPS * 1e12) + alg32 / (PEAK_FP32_TFLOPS * 1e12))
        out["hierarchy_model_flops_per_ray_bounce"] = {"f64": alg64, "f32": alg32}
    return out


def committed_pmc(config, width, height, spp, world, any_spp=False):
    """The committed PMC summary of this configuration (profiles/pmc_c<N>.json, written by
    tools/summarize_pmc_cfg.py from separate rocprofv3 --pmc passes), or None: PMC counters cannot be
    collected from inside the benchmark process.  The record must match the image size and GPU count;
    spp too unless any_spp (the per-configuration array: config 5's committed pass is at 256 spp while the
    array runs its own 4096 -- per-ray figures carry over, per-launch bytes are scaled and say so)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", f"pmc_c{config}.json")))
        if (rec["config"], rec["width"], rec["height"], rec["n_gpus"]) == (config, width, height, world) and \
                (any_spp or rec["spp"] == spp):
            rec["file"] = f"profiles/pmc_c{config}.json"
            return rec
    except Exception:
        pass
    return None


def pmc_keys(pmc, spp, kernel_ms=None):
    """hardware-true figures of a configuration's dominant kernel from its committed PMC pass -- or nulls with
    `pmc_stale` when that pass no longer describes the kernel timed now (pmc_is_stale)"""
    empty = {"valu_busy": None, "lane_utilisation": None, "frac_valu_lanes": None, "valu_instr_per_64_bounces": None, "traffic": None,
             "pmc_source": None}
    if not pmc:
        return empty
    stale = pmc_is_stale(pmc, kernel_ms, spp) if kernel_ms is not None else None
    if stale:
        empty.update({"pmc_stale": True, "pmc_stale_reason": f"{pmc.get('file')}: {stale}"})
        return empty
    same = pmc["spp"] == spp
    src = pmc.get("source", pmc["file"])
    return {"valu_busy": pmc.get("valu_busy"), "lane_utilisation": pmc.get("lane_utilisation"),
            "frac_valu_lanes": (min(pmc["valu_busy"], 1.0) * pmc["lane_utilisation"]) if pmc.get("valu_busy") is not None and
                               pmc.get("lane_utilisation") is not None else None,   # issue busy x lanes active: the hardware-true fraction
            "valu_instr_per_64_bounces": pmc.get("valu_instr_per_64_bounces"),
            # HBM bytes per launch; a pass at another spp is scaled by the sample count (ring / table traffic is per ray)
            "traffic": pmc["traffic_bytes_per_launch"] * (1.0 if same else spp / pmc["spp"]),
            "pmc_stale": False,
            "pmc_source": f"committed PMC pass {src} (rocprofv3 --pmc, separate passes; FETCH_SIZE doubled per the gfx950 "
                          "correction; source hash and kernel time checked against this run)" +
                          ("" if same else f"; from the {pmc['spp']}-spp pass: per-ray figures as measured, "
                                           f"traffic scaled x{spp / pmc['spp']:g} to this launch's {spp} spp")}


def flops_per_ray(n_spheres, n_triangles):
    """SURVEY.md section 8d counting convention (add/sub/mul/div/sqrt = 1, no FMA credit)."""
    return 17.0 * n_spheres + 40.0 * n_triangles + 120.0


def host_cpus():
    """(logical CPUs of the host, CPUs this process may use): the second is what a worker pool
    should be sized by -- the scheduler affinity mask capped by the cgroup CPU quota."""
    nproc = os.cpu_count() or 1
    usable = nproc
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    try:  # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    return nproc, usable, quota


# ---- CPU baseline: the compiled reference on a bounded sample -----------------------------

GLASS_DEPTH = 5   # the reference's own compile-time MAX_DEPTH (raytracer.h:25)


def make_scene(name, width=None, height=None, spp=None, depth=None):
    """A benchmark scene by name -- the GPU side and the CPU workers build the same bytes from it.
      <N>     BASELINE configs[N - 1] (rt_scene_build of the host library)
      glass   config 4's room with every fifth of its 30 packed spheres turned into white M_REFRACTION 'glass' -- the
              share the reference's own generator gives that material (main.c:107-115: r > 0.8) --, MAX_DEPTH 5: what
              trace_path's two-child branch (raytracer.c:514-539) costs; the `integrators` entries of the N = 1 line"""
    from rt_amd import abi, scene as S
    if name == "glass_mesh":
        # config 5's scene with its 10,240-triangle mesh turned M_REFRACTION, MAX_DEPTH 5: the two-child branch through the
        # hierarchy (pt_render_tiles_tri_queued_refr*: parked walks + windowed sums + pending children that travel through the ring)
        sc = S.build_scene(5, width, height, spp, GLASS_DEPTH if depth is None else depth)
        sc.meshes[0].flags = abi.M_REFRACTION
        return sc
    if name != "glass":
        return S.build_scene(int(name), width, height, spp, depth)
    room = S.build_scene(4, width, height, spp)
    objs = []
    for i in range(room.n_objects):
        o = room.objects[i]
        d = dict(flags=int(o.flags), radius=float(o.radius), center=o.center.tuple(), color=o.color.tuple(), emission=o.emission.tuple())
        if i >= 8 and (i - 8) % 5 == 0:
            d.update(flags=abi.M_REFRACTION, color=(1.0, 1.0, 1.0), emission=(0.0, 0.0, 0.0))
        objs.append(d)
    info = S.scene_info(4)
    sc = S.custom_scene(objs, room.width, room.height, room.samples, GLASS_DEPTH if depth is None else depth,
                        tuple(info.cam_pos), tuple(info.cam_target))
    room.free()
    return sc


def _cpu_worker(args):
    name, width, height, spp, depth, pixels, kind, want_pixels, integrator = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    sc = make_scene(name, width, height, spp, depth)
    if kind == "reference":
        # the reference's own compiled trace_path() / cast_ray() / intersect(); scenes with meshes through its revived mesh
        # scan (oracle/ref_harness.c, ORACLE_MESH_HOOK)
        ora = oracle_py.RefMeshOracle(depth) if sc.n_meshes else oracle_py.RefOracle(depth)
        mean, rgb8, st = ora.render_pixels(sc, SEED, pixels=pixels, want_rgb8=want_pixels, integrator=integrator)
        casts = st["tests"] // max(sc.n_primitives, 1)   # its counter is per primitive test: n_primitives per scan
    else:
        mean, rgb8, st = oracle_py.PtOracle().render_pixels(sc, SEED, pixels=pixels, want_rgb8=want_pixels, integrator=integrator)
        casts = st["casts"]
    return casts, st["rays"], (mean, rgb8) if want_pixels else None


def sample_tiles(width, height, budget_tiles):
    """-> (tile_first, tile_stride, tile_count): `budget_tiles` 8x8 tiles spread evenly over the frame as ONE strided
    run, so that the GPU can re-render exactly this subset (render_tiles' tile_first / tile_stride / tile_count)."""
    tx, ty = (width + 7) // 8, (height + 7) // 8
    total = tx * ty
    count = max(1, min(budget_tiles, total))
    stride = max(total // count, 1)
    first = (total - 1 - stride * (count - 1)) // 2
    return first, stride, count


def tile_pixel_indices(width, height, first, stride, count, tiles=None):
    """-> (linear pixel indices [n], slot in the compact tile buffer [n], pixel-in-tile [n]) of the inside-image pixels of
    tiles first, first + stride, ... (or of the explicit list `tiles`): tile by tile, row-major inside a tile"""
    import numpy as np
    tx = (width + 7) // 8
    px, slot, pit = [], [], []
    ids = [first + k * stride for k in range(count)] if tiles is None else [int(t) for t in tiles]
    for k, t in enumerate(ids):
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        for r in range(8):
            for c in range(8):
                if x0 + c < width and y0 + r < height:
                    px.append((y0 + r) * width + x0 + c)
                    slot.append(k)
                    pit.append(r * 8 + c)
    return np.array(px, dtype=np.uint32), np.array(slot, dtype=np.int64), np.array(pit, dtype=np.int64)


def cpu_reference(config, width, height, spp, depth, n_meshes, budget_tiles, workers, want_pixels=True, integrator="path"):
    """The CPU checker on `budget_tiles` 8x8 tiles spread evenly over the frame at full spp, one single-threaded
    process per usable core (processes, not threads: the reference's global counters make its threads slower,
    SURVEY T7).  -> (cpu_baseline dict, sample dict): the timing, and -- since round 4 -- the pixels it rendered
    (linear fp64 means, tonemapped bytes, rays, casts) for the same-run parity check against the GPU frame."""
    import multiprocessing as mp
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    have_ref = oracle_py.ref_mesh_available(depth) if n_meshes else oracle_py.ref_available(depth)
    kind = "reference" if have_ref else "port"
    tile_list = budget_tiles["tiles"] if isinstance(budget_tiles, dict) else None    # an explicit list of tiles
    if tile_list is not None:
        first, stride, count = -1, 0, len(tile_list)
    else:
        first, stride, count = budget_tiles if isinstance(budget_tiles, tuple) else sample_tiles(width, height, budget_tiles)
    total = ((width + 7) // 8) * ((height + 7) // 8)
    px, _, _ = tile_pixel_indices(width, height, first, stride, count, tiles=tile_list)
    nproc, usable, quota = host_cpus()
    cores = max(1, min(workers or usable, len(px)))
    chunks = [px[i::cores] for i in range(cores)]  # interleaved: even load
    jobs = [(str(config), width, height, spp, depth, ch, kind, want_pixels, integrator) for ch in chunks]
    ctx = mp.get_context("spawn")
    with ctx.Pool(len(jobs)) as pool:
        pool.map(_cpu_worker, [(str(config), width, height, 1, depth, ch[:1], kind, False, integrator) for ch in chunks])  # warm
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, jobs)
        dt = time.perf_counter() - t0
    casts = sum(r[0] for r in res)
    rays = sum(r[1] for r in res)
    oracle_name = ("oracle/_ref (the reference's compiled trace_path / intersect" +
                   ("; meshes through its revived mesh scan, ref_harness.c ORACLE_MESH_HOOK)" if n_meshes else ")")
                   if kind == "reference" else "oracle/pt_oracle.c (CPU restatement, bit-pinned to oracle/_ref by tests/test_oracle_ref.py)")
    baseline = {"value": casts / dt, "unit": "ray-bounces/s", "cores": len(jobs), "kind": kind,
                "host_nproc": nproc, "host_usable_cpus": usable, "host_cgroup_cpu_quota": quota,
                "sample": f"{count} of {total} 8x8 tiles ({'an explicit list' if tile_list is not None else f'tile {first} + {stride} k'}; {len(px)} pixels) at {spp} spp, "
                          f"{len(jobs)} single-thread processes (host: {nproc} logical CPUs, {usable} usable by this "
                          f"process), {dt:.1f} s wall",
                "mpixel_samples_per_s": len(px) * spp / dt * 1e-6}
    sample = {"first": first, "stride": stride, "count": count, "px": px, "rays": rays, "casts": casts, "spp": spp,
              "oracle": oracle_name, "seconds": dt, "tiles": tile_list}
    if want_pixels:
        mean = np.zeros((len(px), 3))
        rgb8 = np.zeros((len(px), 3), dtype=np.uint8)
        for i, r in enumerate(res):      # undo the interleaving
            mean[i::cores] = r[2][0]
            rgb8[i::cores] = r[2][1]
        sample["mean"], sample["rgb8"] = mean, rgb8
    return baseline, sample


RMS_BAR = 1e-4   # BASELINE.json north_star: per-channel RMS of the linear float framebuffer against the reference's


def parity_numbers(gpu_rgb, gpu_rgb8, gpu_rays, gpu_casts, sample, hdr=False):
    """Pure comparison (numpy in, dict out; tests/test_host.py covers it on the CPU): the GPU's linear floats and
    tonemapped bytes of the sample's pixels against the CPU reference's, and the decision-exactness counters."""
    import numpy as np
    g = np.asarray(gpu_rgb, dtype=np.float64).reshape(-1, 3)
    m = np.asarray(sample["mean"], dtype=np.float64).reshape(-1, 3)
    d = g - m
    rms = np.sqrt((d * d).mean(axis=0)) if len(d) else np.zeros(3)
    d8 = np.abs(np.asarray(gpu_rgb8, dtype=np.int16).reshape(-1, 3) - np.asarray(sample["rgb8"], dtype=np.int16).reshape(-1, 3))
    counters_equal = int(gpu_rays) == int(sample["rays"]) and int(gpu_casts) == int(sample["casts"])
    finite = bool(np.isfinite(g).all() and np.isfinite(m).all())
    out = {"pixels": int(len(m)), "tiles": int(sample["count"]), "tile_first": int(sample["first"]),
           "tile_stride": int(sample["stride"]), "spp": int(sample["spp"]),
           "rms": [float(v) for v in rms], "max_abs": float(np.abs(d).max()) if len(d) else 0.0,
           "u8_max_diff": int(d8.max()) if len(d8) else 0, "counters_equal": bool(counters_equal),
           "rays": {"gpu": int(gpu_rays), "cpu": int(sample["rays"])},
           "ray_bounces": {"gpu": int(gpu_casts), "cpu": int(sample["casts"])},
           "oracle": sample["oracle"], "cpu_seconds": float(sample["seconds"]),
           "bar": f"per-channel RMS <= {RMS_BAR:g} (linear float framebuffer), tonemapped bytes within 1 LSB, rays and "
                  "ray-bounces of the pixel subset equal"}
    # hdr (the glass scene only: the reference's "fresnel" weight reaches 7.3 per M_REFRACTION hit from inside a sphere, so
    # pixel values are not bounded by the emitters'): the float32 framebuffer holds 6e-8 RELATIVE, the bar scales with the
    # largest value compared -- the same rule as the tests' assert_parity(hdr=True)
    bar = RMS_BAR * (max(1.0, float(np.abs(m).max())) if (hdr and len(m)) else 1.0)
    if hdr:
        out["bar"] += f"; HDR scene: RMS bar scaled by the largest value compared -> {bar:.3g}"
    out["ok"] = bool(finite and (rms <= bar).all() and out["u8_max_diff"] <= 1 and counters_equal)
    if hdr:   # (what the north star's own absolute bar says, scaled or not)
        out["passes_unscaled_bar"] = bool(finite and (rms <= RMS_BAR).all())
    return out


def gpu_parity(gs, sc, dev, sample, frame, frame8, integrator="path", hdr=False):
    """The same-run parity block: `frame` / `frame8` are the row-major images the TIMED steps produced (device tensors);
    their pixels of the sample's tiles are compared with the CPU reference's, and the subset is rendered once more on
    the GPU (render_tiles with tile_first / tile_stride: outside the timed region) for its own ray / ray-bounce
    counters -- which must equal the reference's -- and to show that the timed frame holds exactly those values."""
    import numpy as np
    import torch
    from rt_amd import abi
    first, stride, count = sample["first"], sample["stride"], sample["count"]
    px, slot, pit = tile_pixel_indices(sc.width, sc.height, first, stride, count)
    assert (px == sample["px"]).all()
    st = torch.zeros(abi.NSTATS, dtype=torch.int64, device=dev)
    t, t8, _ = gs.render_tiles(SEED, first, stride, count, stats=st, integrator=integrator)
    torch.cuda.synchronize(dev)
    st = st.cpu().tolist()
    sub = t.cpu().numpy()[slot, pit]
    sub8 = t8.cpu().numpy()[slot, pit]
    idx = torch.from_numpy(px.astype(np.int64)).to(dev)
    fr = frame.reshape(-1, 3)[idx].cpu().numpy()
    fr8 = frame8.reshape(-1, 3)[idx].cpu().numpy()
    out = parity_numbers(fr, fr8, st[abi.STAT_RAYS], st[abi.STAT_CASTS], sample, hdr=hdr)
    out["timed_frame_equals_rerender"] = bool((fr.view(np.uint32) == sub.view(np.uint32)).all() and (fr8 == sub8).all())
    out["ok"] = bool(out["ok"] and out["timed_frame_equals_rerender"])
    out["compared"] = ("the frame the timed steps produced, at the sample's pixels, against the CPU reference's means; "
                       "counters from a re-render of exactly those tiles")
    return out


def config5_wide_parity(gs, sc, dev, cpu_workers, spp=4, n_sil=128, n_in=128, n_out=768):
    """The second parity sample of BASELINE configs[4] AT ITS OWN 3840x2160: 1,024 tiles = 65,536 pixels at
    `spp` samples against the reference's compiled code with its mesh scan revived (RefMeshOracle: 10,248 tests per scan) --
    128 tiles that straddle the mesh's outline, 128 inside it, 768 strided over the rest of the frame, which see the mesh only
    through a bounce (rt_amd.scene.mesh_view_tiles).  What this pins at full resolution and the two 4096-spp tiles cannot: the
    resolution-dependent conservative rules of the hierarchy kernels (a tile's cone of camera rays against the mesh's
    bounding ball, the probe that lets most camera rays skip the walk) across the frame.  Each tile is rendered on its own
    (render_tiles, tile_first = t, count 1, samples = spp): pixels, bytes, rays and ray-bounces of the subset."""
    import numpy as np
    import torch
    from rt_amd import abi, scene as S
    view = S.mesh_view_tiles(sc)
    tiles = np.concatenate([S.pick_evenly(view["silhouette"], n_sil), S.pick_evenly(view["inside"], n_in),
                            S.pick_evenly(view["outside"], n_out)]).astype(np.uint32)
    _, sample = cpu_reference(sc.config, sc.width, sc.height, spp, sc.max_depth, sc.n_meshes, {"tiles": tiles}, cpu_workers)
    px, slot, pit = tile_pixel_indices(sc.width, sc.height, -1, 0, len(tiles), tiles=tiles)
    assert (px == sample["px"]).all()
    t_all = torch.empty((len(tiles), abi.TILE_PIXELS, 3), dtype=torch.float32, device=dev)
    t8_all = torch.empty((len(tiles), abi.TILE_PIXELS, 3), dtype=torch.uint8, device=dev)
    st = torch.zeros(abi.NSTATS, dtype=torch.int64, device=dev)
    for k, t in enumerate(tiles):
        gs.render_tiles(SEED, int(t), 1, 1, t_all[k:k + 1], t8_all[k:k + 1], st, samples=spp)
    torch.cuda.synchronize(dev)
    gs.launch_status()
    st = st.cpu().tolist()
    out = parity_numbers(t_all.cpu().numpy()[slot, pit], t8_all.cpu().numpy()[slot, pit], st[abi.STAT_RAYS], st[abi.STAT_CASTS], sample)
    out["tiles_by_view"] = {"straddle_the_mesh_outline": int(min(n_sil, len(view["silhouette"]))),
                            "inside_the_outline": int(min(n_in, len(view["inside"]))),
                            "see_the_mesh_only_through_a_bounce": int(min(n_out, len(view["outside"])))}
    out["kernel"] = gs.last_launch_kernel()
    out["compared"] = (f"{len(tiles)} tiles of the {sc.width}x{sc.height} frame, each rendered on its own at {spp} spp, against the "
                       "CPU reference's means of the same pixels and samples")
    return out


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
    threads = host_cpus()[1]
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


# ---- the other configurations, one GPU -----------------------------------------------------

def isa_keys(kernel, source_sha=None):
    """VGPRs / SGPRs / scratch bytes / LDS of a kernel from the committed ISA statistics (profiles/isa_stats.json, written by
    tools/isa_stats.sh on the cross-compiler: no GPU needed) -- only while they describe today's sources"""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "isa_stats.json")))
    except Exception:
        return None
    want = source_sha if source_sha is not None else kernel_source_sha256()
    if rec.get("source_sha256") != want:
        return {"isa_stale": True}
    k = rec["kernels"].get(kernel)
    return dict(k, source="profiles/isa_stats.json (tools/isa_stats.sh)") if k else None


def integrator_line(kind, dev, parity_tiles=0, cpu_workers=0):
    """The two other code paths of render() that the reference ships next to the headline's, on this GPU,
    1920x1080:  'glass' -- trace_path with M_REFRACTION's two children per hit (raytracer.c:514-539), make_scene('glass'),
    64 spp;  'cast_ray' -- the Whitted integrator on the other side of render()'s `#if 1` (raytracer.c:207-211, :556-641) on
    config 4's room, 16 spp.  Kernel time by HIP events over 3 frames, the kernel's registers / scratch from the committed
    ISA statistics, and the same-run parity block against the reference's compiled code."""
    import torch
    from rt_amd import abi, gpu as G
    name, spp, integrator = {"glass": ("glass", 64, "path"), "glass_mesh": ("glass_mesh", 16, "path")}.get(kind, ("4", 16, "whitted"))
    sc = make_scene(name, None, None, spp)
    gs = G.GpuScene(sc, device=dev.index)
    total = G.n_tiles(sc.width, sc.height)
    tiles = torch.empty((total, abi.TILE_PIXELS, 3), dtype=torch.float32, device=dev)
    tiles8 = torch.empty((total, abi.TILE_PIXELS, 3), dtype=torch.uint8, device=dev)
    stats = torch.zeros(abi.NSTATS, dtype=torch.int64, device=dev)
    gs.render_tiles(SEED, 0, 1, total, tiles, tiles8, stats, integrator=integrator)
    torch.cuda.synchronize(dev)
    stats.zero_()
    steps = 3
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        gs.render_tiles(SEED, 0, 1, total, tiles, tiles8, stats, integrator=integrator)
        b.record()
    torch.cuda.synchronize(dev)
    ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    rays, casts, tests, samples = [int(v) / steps for v in stats.cpu().tolist()]
    kernel = gs.last_launch_kernel()   # what the timed launches took (the launch's own facts included), not only what the scene suggests
    what = {"glass": ("trace_path, M_REFRACTION scene (raytracer.c:514-539)", "config 4 room, every fifth packed sphere M_REFRACTION"),
            "glass_mesh": ("trace_path, M_REFRACTION mesh through the hierarchy (raytracer.c:514-539)",
                           "BASELINE configs[4] scene with its mesh turned M_REFRACTION")}.get(kind, ("cast_ray (raytracer.c:556-641)", "BASELINE configs[3] scene"))
    line = {"integrator": what[0],
            "workload": f"{what[1]}: {sc.width}x{sc.height}, {sc.samples} spp, {sc.n_objects} spheres + {sc.n_triangles} triangles, depth {sc.max_depth}",
            "kernel": kernel, "kernel_ms": ms, "steps": steps,
            "ray_bounces_per_s": casts / (ms * 1e-3), "scene_scans_note": "ray-bounces = scene scans (cast_ray: primary + shadow rays)",
            "mpixel_samples_per_s": samples / (ms * 1e-3) * 1e-6, "rays_per_sample": rays / max(samples, 1),
            "isa": isa_keys(kernel)}
    if parity_tiles:
        try:
            image, image8 = gs.untile(tiles, tiles8, 0, 1, total)
            # (the glass mesh: 64 tiles around the frame's centre, on the mesh -- 10,248 primitive tests per scan on the CPU)
            budget = (135 * 480 + 208, 1, 64) if kind == "glass_mesh" else parity_tiles
            _, sample = cpu_reference(name, sc.width, sc.height, sc.samples, sc.max_depth, sc.n_meshes, budget, cpu_workers,
                                      integrator=integrator)
            line["parity"] = gpu_parity(gs, sc, dev, sample, image, image8, integrator=integrator, hdr=(kind in ("glass", "glass_mesh")))
        except Exception as exc:
            line["parity"] = {"ok": False, "error": repr(exc)}
    gs.close()
    return line


# 8x8 tiles of the per-configuration parity samples, at the configuration's own size and spp: a budget (spread evenly over
# the frame) or an explicit (tile_first, tile_stride, tile_count).  Config 5 is 4096 spp x 10,248 primitives per scan on the
# CPU -- ~8 s of 16 cores per tile --, so two tiles: the centre of the frame (on the mesh) and one on the floor
PARITY_TILES = {1: 1024, 2: 1024, 3: 1024, 5: (135 * 480 + 240, 60 * 480 - 150, 2)}


def config_line(cfg, spp, steps, dev, parity_tiles=0, cpu_workers=0):  # parity_tiles: 0 = no parity block
    """One BASELINE configuration on this GPU: `steps` full frames, kernel time by HIP events; with
    parity_tiles > 0 also the same-run parity block against the compiled reference (gpu_parity)."""
    import torch
    from rt_amd import abi, gpu as G, scene as S
    sc = S.build_scene(cfg, samples=spp or None)
    gs = G.GpuScene(sc, device=dev.index)
    total = G.n_tiles(sc.width, sc.height)
    chunks = gs.suggest_chunks(total)
    tiles = torch.empty((total, abi.TILE_PIXELS, 3), dtype=torch.float32, device=dev)
    tiles8 = torch.empty((total, abi.TILE_PIXELS, 3), dtype=torch.uint8, device=dev)
    stats = torch.zeros(abi.NSTATS, dtype=torch.int64, device=dev)
    for _ in range(1 if steps <= 3 else 5):                                      # warm (the short frames: as the headline, 5)
        gs.render_tiles(SEED, 0, 1, total, tiles, tiles8, stats, chunks=chunks)
    torch.cuda.synchronize(dev)
    stats.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        gs.render_tiles(SEED, 0, 1, total, tiles, tiles8, stats, chunks=chunks)
        b.record()
    torch.cuda.synchronize(dev)
    ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    rays, casts, tests, samples = [int(v) / steps for v in stats.cpu().tolist()]
    fr = flops_per_ray(sc.n_objects, sc.n_triangles)
    line = {"config": cfg, "workload": f"BASELINE configs[{cfg - 1}]: {sc.width}x{sc.height}, {sc.samples} spp, "
                                       f"{sc.n_objects} spheres + {sc.n_triangles} triangles, depth {sc.max_depth}",
            "kernel": gs.last_launch_kernel(), "kernel_ms": ms, "steps": steps, "ray_bounces_per_s": casts / (ms * 1e-3),
            "mpixel_samples_per_s": samples / (ms * 1e-3) * 1e-6, "rays_per_sample": rays / max(samples, 1),
            "flops_per_ray_bounce": fr, "frac": casts * fr / (ms * 1e-3) * 1e-12 / PEAK_FP64_TFLOPS, "isa": isa_keys(gs.last_launch_kernel())}
    line.update(pmc_keys(committed_pmc(cfg, sc.width, sc.height, sc.samples, 1, any_spp=True), sc.samples, ms))
    line.update(executed_work(committed_diag(cfg), casts, ms * 1e-3, sc.n_objects, sc.n_triangles))
    if parity_tiles:
        try:
            image, image8 = gs.untile(tiles, tiles8, 0, 1, total)     # the last timed frame, row-major
            _, sample = cpu_reference(cfg, sc.width, sc.height, sc.samples, sc.max_depth, sc.n_meshes, parity_tiles, cpu_workers)
            line["parity"] = gpu_parity(gs, sc, dev, sample, image, image8)
            if cfg == 5 and sc.width >= 1920:
                # ... and the wide sample at the frame's own resolution: >= 65k pixels, few samples each
                wide = config5_wide_parity(gs, sc, dev, cpu_workers)
                line["parity"]["wide"] = wide
                line["parity"]["ok"] = bool(line["parity"]["ok"] and wide["ok"])
        except Exception as exc:
            line["parity"] = {"ok": False, "error": repr(exc)}
    nominal = S.scene_info(cfg).samples
    if sc.samples != nominal:
        line["note"] = f"reduced spp: the configuration's own is {nominal}"
    if sc.n_triangles > 256:
        # the hierarchy skips nearly all of the model's O(N) triangle tests: the SURVEY-8d fraction (40 flops for each of
        # the scene's triangles per ray-bounce) means nothing here.  `frac` is then the same model with the hierarchy in
        # place of the O(N) triangle scan -- spheres and shading as SURVEY 8d counts them, the triangles by the node
        # visits, leaf pre-tests and exact tests the walk executes (executed_work: frac_hierarchy_model)
        line["frac_scan_model"] = line["frac"]
        line["frac"] = line.get("frac_hierarchy_model")
        line["note"] = (line.get("note", "") + "; " if "note" in line else "") + \
            ("frac = the algorithmic model with the hierarchy's executed node visits / pre-tests / exact tests in place of "
             "the reference's O(N) triangle scan (the O(N) form, frac_scan_model, is > 1 and means nothing: the hierarchy "
             "legitimately skips almost all of those tests)")
    if ms < 0.3:
        line["note"] = (line.get("note", "") + "; " if "note" in line else "") + "launch-bound at this size"
    gs.close()
    sc.free()
    return line


# ---- the single-process C path: rt_hip_render_image over N devices ---------------------------

def host_path_main(args):
    """Child mode (--host-path): ONE process drives N devices through rt_hip_render_image()
    (grouped RCCL send/recv to device 0 inside the shim): what the C host's render() does."""
    import torch  # noqa: F401  (one HIP runtime: load it before the shim)
    from rt_amd import abi, gpu as G, scene as S
    import ctypes as C
    shim = abi.load_shim()
    have = shim.rt_hip_device_count()
    n = args.gpus
    logical = args.logical_devices
    if logical:
        # N LOGICAL devices on this box's GPU 0 (rt_hip_set_device_map): the whole n_devices > 1 path of the C host -- partition,
        # per-device launches, the gather's slot arithmetic, the per-segment scatter, the counter sums -- executed on one GPU
        m = (C.c_int * n)(*([0] * n))
        if shim.rt_hip_set_device_map(m, n) != 0:
            print(json.dumps({"host_path": {"error": shim.rt_hip_last_error().decode()}}))
            return 0
    elif have < n:
        print(json.dumps({"host_path": {"error": f"{have} devices visible, {n} requested"}}))
        return 0
    sc = S.build_scene(args.config, args.width or None, args.height or None, args.spp or None)
    out = {"n_devices": n, "workload": f"{sc.width}x{sc.height}, {sc.samples} spp"}
    if logical:
        out["logical_devices_on_one_gpu"] = True
        out["note"] = (f"{n} logical devices mapped onto GPU 0: the C host's multi-device code path executed and checked on one "
                       "GPU; its time is NOT a scaling measurement (the devices share the GPU)")
    times, kernel_s, phases = [], [], []
    ref_img = None
    ph = (C.c_double * 3)()
    for k in range(args.warmup + args.steps):
        t0 = time.perf_counter()
        img, img8, st, secs = G.render_image_host(sc, SEED, n_devices=n)
        dt = time.perf_counter() - t0
        shim.rt_hip_last_image_phases(ph)
        if k == 0:
            out["first_call_ms"] = dt * 1e3   # context build included: scene upload, buffers, workspaces (communicators for N > 1)
            out["first_call_context_ms"] = ph[0] * 1e3
        if k >= args.warmup:
            times.append(dt)
            kernel_s.append(secs)
            phases.append([ph[0] * 1e3, ph[1] * 1e3, ph[2] * 1e3])
        if ref_img is None:
            ref_img = img
    out["call_ms"] = [t * 1e3 for t in times]           # the whole call: scene compare, launches, kernels, gather, the frame over PCIe
    out["kernel_ms"] = [t * 1e3 for t in kernel_s]      # render kernels, max over devices
    out["phase_ms"] = {"context": min(p[0] for p in phases), "launch_to_idle": min(p[1] for p in phases), "copy_out": min(p[2] for p in phases),
                       "note": "host clock inside rt_hip_render_image: [context] scene compare (a rebuild on the first call only), "
                               "[launch_to_idle] launches + kernels + gather + scatter until every stream is idle, [copy_out] frame, "
                               "bytes and counters over PCIe"}
    out["ray_bounces_per_s"] = st["casts"] / min(times)
    out["kernel"] = shim.rt_hip_last_launch_kernel().decode()
    if n > 1:  # the assembled frame must equal the one-device frame bit for bit
        one, one8, st1, _ = G.render_image_host(sc, SEED, n_devices=1)
        out["equals_one_device_frame"] = bool((one == ref_img).all()) and bool((one8 == img8).all()) and st1 == st
    print(json.dumps({"host_path": out}), flush=True)
    return 0


def run_host_path_child(args, n, timeout_s=240, logical=False):
    cmd = [sys.executable, os.path.abspath(__file__), "--host-path", "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--config", str(args.config), "--spp", str(args.spp), "--width", str(args.width), "--height", str(args.height)]
    if logical:
        cmd.append("--logical-devices")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        for ln in reversed(p.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)["host_path"]
        return {"error": f"rc {p.returncode}: {(p.stderr or p.stdout)[-400:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s} s"}
    except Exception as exc:
        return {"error": repr(exc)}


def run_cli_host(args, timeout_s=300):
    """The C command-line host (raytracer.c_amd/host/raytracer = the reference's main.c flags) as a process: wall time from
    exec to exit for the bench's workload, and where it went -- the host prints its own phase clock (HIP runtime start,
    context = scene upload + buffers + workspaces, render, frame over PCIe, PNG encode + write)."""
    exe = os.path.join(ROOT, "raytracer.c_amd", "host", "raytracer")
    if not os.path.exists(exe):
        return {"error": "raytracer.c_amd/host/raytracer is not built"}
    import tempfile
    from rt_amd import scene as S
    info = S.scene_info(args.config)
    w, h, spp = args.width or info.width, args.height or info.height, args.spp or info.samples
    with tempfile.TemporaryDirectory() as tmp:
        out_png = os.path.join(tmp, "bench_cli.png")
        cmd = [exe, "-w", str(w), "-h", str(h), "-s", str(spp), "-c", str(args.config), "-d", str(info.max_depth), "-o", out_png]
        try:
            t0 = time.perf_counter()
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            wall = time.perf_counter() - t0
        except subprocess.TimeoutExpired:
            return {"error": f"timed out after {timeout_s} s"}
        if p.returncode != 0:
            return {"error": f"rc {p.returncode}: {(p.stderr or p.stdout)[-300:]}"}
        res = {"command": " ".join(cmd[:-1] + ["<tmp>.png"]), "process_wall_ms": wall * 1e3, "png_bytes": os.path.getsize(out_png)}
        for ln in p.stdout.splitlines():
            if ln.startswith("phases: "):
                for part in ln[len("phases: "):].split(", "):
                    k, v = part.rsplit(" ", 2)[0], part.rsplit(" ", 2)[1]
                    res.setdefault("phase_ms", {})[k.replace(" ", "_")] = float(v) * 1e3
            if ln.startswith("cast "):
                res["rays"] = int(ln.split()[1])
        return res


# ---- `python bench.py --gpus N` without a launcher: start one rank per GPU ourselves ---------

def launch_ranks(n):
    """Called as plain `python bench.py --gpus N` (N > 1, no RANK in the environment): run the same
    command line under torch.distributed.run as a CHILD process -- one rank per GPU, rendezvous on
    127.0.0.1 -- relay rank 0's JSON line and return the child's exit code.  This parent may initialise
    the HIP runtime (torch.cuda.device_count() falls back to hipGetDeviceCount where amdsmi is absent), which
    is harmless because it only ever STARTS a fresh child (subprocess) and never replaces itself: replacing a
    process that has initialised HIP takes the machine down on this pool, so nothing here execs.
    HSA_ENABLE_IPC_MODE_LEGACY: the pool's host driver supports dmabuf IPC only, RCCL between processes fails
    with `hipIpcGetMemHandle: invalid argument` without the value 0; the task environment exports it, and an
    external launcher (torchrun started by the driver) inherits it the same way -- it is set here only when the
    environment does not have it at all, never overridden."""
    import socket
    with socket.socket() as s:               # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    try:
        import torch
        have = torch.cuda.device_count()
    except Exception:
        have = 0
    if have < n and not rehearse:
        print(json.dumps({"metric": "ray-bounces/sec", "value": None, "unit": "ray-bounces/s", "n_gpus": n,
                          "error": f"--gpus {n} but {have} GPU(s) visible on this node (RT_BENCH_REHEARSE=1 runs the "
                                   f"{n}-rank control flow on one GPU over gloo; its number is not a measurement)"}), flush=True)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)     # stderr passes through
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)      # anything else the ranks printed is not the result line
    if lines:
        print(lines[-1], flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks printed no JSON line", file=sys.stderr)
        return 4
    return proc.returncode


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
    ap.add_argument("--cpu-tiles", type=int, default=2048, help="8x8 tiles of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="processes of the CPU baseline (0 = the CPUs this process may use)")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration array")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the same-run parity blocks (the CPU baseline is still timed; its pixels are thrown away)")
    ap.add_argument("--c5-spp", type=int, default=0,
                    help="spp of config 5 in the per-configuration array (0 = its own 4096: ~4 s a frame on one GPU)")
    ap.add_argument("--integrators-only", action="store_true",
                    help="dev: only the `integrators` entries (glass scene, cast_ray), one JSON line, no CPU legs")
    ap.add_argument("--logical-devices", action="store_true",
                    help="with --host-path: --gpus N LOGICAL devices mapped onto GPU 0 (rt_hip_set_device_map)")
    ap.add_argument("--host-path", action="store_true",
                    help="single process: time rt_hip_render_image() over --gpus devices (the C host's path) and exit")
    args = ap.parse_args()
    if args.host_path:
        return host_path_main(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args.gpus)   # plain `python bench.py --gpus N`: start the ranks ourselves
    if args.integrators_only:
        import torch
        dev = torch.device("cuda", 0)
        print(json.dumps({"integrators": [integrator_line(k, dev) for k in ("glass", "glass_mesh", "cast_ray")]}), flush=True)
        return 0

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
    # a CPU-side group for waits during which the GPUs must stay idle (an RCCL barrier is a spinning kernel)
    ctrl = dist.new_group(backend="gloo") if world > 1 else None

    # how many ranks the backend really connected: an all-reduce of ones
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.int32, device="cpu" if rehearse else dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())

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
    phase_events = []   # per timed step: (render start, render end, gather end, untile end)

    def step(timed):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()                                 # same stream the shim launches on
        gs.render_tiles(SEED, first, stride, count, tiles, tiles8, stats, chunks=chunks, workspace=workspace)
        ev[1].record()
        # RCCL over xGMI when world > 1: the one exchange of the path
        parts, parts8 = D.gather_tiles(tiles, tiles8, rank, world, gathered, gathered8, via_cpu=rehearse)
        ev[2].record()
        if rank == 0:
            for r, f, s_, c in D.segments(W, H, world):
                gs.untile(parts[r], parts8[r], f, s_, c, image, image8)
        ev[3].record()
        if timed:
            phase_events.append(ev)

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

    n_ev = max(len(phase_events), 1)
    phases = [sum(ev[k].elapsed_time(ev[k + 1]) for ev in phase_events) / n_ev for k in range(3)]  # render, gather, untile
    cdev = "cpu" if rehearse else dev
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    ph = torch.tensor(phases, dtype=torch.float64, device=cdev)
    tot = stats.clone().to(cdev)
    per_rank = [ph.clone() for _ in range(world)] if world > 1 else [ph]
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank, ph)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rank_phase = [[float(v) for v in p.cpu().tolist()] for p in per_rank]
    kern_s = max(p[0] for p in rank_phase) * 1e-3
    rays, casts, tests, samples = [int(v) for v in tot.cpu().tolist()]

    rc = 0
    if rank == 0:
        steps = max(args.steps, 1)
        casts_per_step = casts / steps
        fr = flops_per_ray(sc.n_objects, sc.n_triangles)
        # dominant kernel; its work per launch on the slowest rank ~ 1/world of the frame
        launch_casts = casts_per_step / world
        achieved_tflops = launch_casts * fr / kern_s * 1e-12 if kern_s > 0 else 0.0
        alg_bytes = (12 + 3) * W * H / world + 88 * sc.n_objects + 72 * sc.n_triangles
        pmc = committed_pmc(args.config, W, H, spp, world)
        pmc_stale = pmc_is_stale(pmc, kern_s * 1e3, spp) if pmc else None
        pmc_stale_reason = f"{pmc['file']}: {pmc_stale}" if pmc_stale else None
        if pmc_stale:
            pmc = None      # a PMC pass that does not describe the kernel timed now is not reported
        pmc_source = (f"committed PMC pass ({pmc.get('source', pmc['file'])}; rocprofv3 --pmc, separate passes; "
                      "FETCH_SIZE doubled per the gfx950 correction; source hash and kernel time checked against this run)") if pmc else None
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
                         "flops": FLOPS_NOTE, "flops_per_ray_bounce": fr, "kernel_ms": kern_s * 1e3, "isa": isa_keys(gs.kernel_name()),
                         "traffic": pmc["traffic_bytes_per_launch"] if pmc else None, "traffic_source": pmc_source,
                         "valu_busy": min(pmc["valu_busy"], 1.0) if pmc and pmc.get("valu_busy") is not None else None,
                         "valu_busy_raw": pmc.get("valu_busy_raw", pmc.get("valu_busy")) if pmc else None,
                         "valu_busy_note": "per-wave quad-cycles over the cycles of the same PMC pass; overlapping waves on a SIMD "
                                           "can push the raw ratio past 1, so valu_busy is capped at 1 (= VALU issue saturated)",
                         "lane_utilisation": pmc.get("lane_utilisation") if pmc else None,
                         # the hardware-true fraction: the share of the VALU's lane-slots that did work (issue busy x lanes
                         # active), whatever the instructions were -- fp64, packed fp32, the RNG's integer ops, selects
                         "frac_valu_lanes": (min(pmc["valu_busy"], 1.0) * pmc["lane_utilisation"]) if pmc and pmc.get("valu_busy") is not None
                                            and pmc.get("lane_utilisation") is not None else None,
                         "valu_instr_per_64_bounces": pmc.get("valu_instr_per_64_bounces") if pmc else None,
                         "source": pmc_source, "pmc_stale": bool(pmc_stale), "pmc_stale_reason": pmc_stale_reason,
                         "note": "branchy fp64 scalar-per-lane math: neither HBM nor MFMA binds it "
                                 "(BASELINE.md section 4); kernel_ms by HIP events on the launch stream"},
            "roofline_hbm": {"bound": "hbm", "kernel": gs.kernel_name(),
                             "achieved": alg_bytes / kern_s * 1e-9 if kern_s > 0 else 0.0, "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": (alg_bytes / kern_s * 1e-9 / PEAK_HBM_GBS) if kern_s > 0 else 0.0,
                             "traffic": pmc["traffic_bytes_per_launch"] if pmc else None, "traffic_source": pmc_source,
                             "algorithmic_bytes_per_launch": alg_bytes},
        }
        out["roofline"].update(executed_work(committed_diag(args.config), launch_casts, kern_s, sc.n_objects, sc.n_triangles))
        if world > 1:
            # the frame assembled from N ranks' tiles must hold the bits one GPU renders: rank 0 renders a strided sample of
            # the frame's tiles alone (outside the timed region) and compares them with the gathered image, bit for bit
            try:
                import numpy as np
                a_first, a_stride, a_count = sample_tiles(W, H, 2048)
                a_px, a_slot, a_pit = tile_pixel_indices(W, H, a_first, a_stride, a_count)
                a_t, a_t8, _ = gs.render_tiles(SEED, a_first, a_stride, a_count)
                torch.cuda.synchronize(dev)
                idx = torch.from_numpy(a_px.astype(np.int64)).to(dev)
                same = bool((image.reshape(-1, 3)[idx].cpu().numpy().view(np.uint32) == a_t.cpu().numpy()[a_slot, a_pit].view(np.uint32)).all() and
                            (image8.reshape(-1, 3)[idx].cpu().numpy() == a_t8.cpu().numpy()[a_slot, a_pit]).all())
                out["assembly_check"] = {"tiles": int(a_count), "pixels": int(len(a_px)), "bit_identical_to_one_gpu": same,
                                         "what": f"the frame gathered from {world} ranks against rank 0's own render of tiles "
                                                 f"{a_first} + {a_stride} k"}
                if not same:
                    out["error"] = "the frame assembled from the ranks' tiles differs from a one-GPU render of the same tiles"
                    rc = 6
            except Exception as exc:
                out["assembly_check"] = {"error": repr(exc)}
            out["ranks_seen"] = ranks_seen
            out["phase_ms"] = {"render": max(p[0] for p in rank_phase), "gather": max(p[1] for p in rank_phase),
                               "untile": rank_phase[0][2],
                               "note": "HIP events per rank around each phase, averaged over the timed steps; render / "
                                       "gather = max over ranks (a rank's gather includes waiting for the slowest "
                                       "renderer), untile = rank 0"}
            out["rank_kernel_ms"] = {"min": min(p[0] for p in rank_phase), "max": max(p[0] for p in rank_phase),
                                     "per_rank": [p[0] for p in rank_phase]}
            if ranks_seen != args.gpus:
                out["error"] = f"the backend connected {ranks_seen} ranks, --gpus asked for {args.gpus}" + ("; " + out["error"] if "error" in out else "")
                rc = 3
        if world == 1 and not args.no_configs and not args.shard:
            lines = []
            for cfg, cspp in ((1, 0), (2, 0), (3, 0), (5, args.c5_spp)):
                if cfg == args.config:
                    continue
                try:
                    # config 5 at its own 4096 spp is ~4 s a frame: one warm-up frame + one timed
                    # frames per configuration: one of config 5's 3.4 s, three of config 3's 19 ms; the millisecond-sized
                    # configurations 1 and 2 get the headline's 20 (+ 5 warm), or the average is the clock's ramp
                    lines.append(config_line(cfg, cspp, 1 if cfg == 5 and cspp in (0, 4096) else (20 if cfg in (1, 2) else 3), dev,
                                             parity_tiles=0 if args.no_parity or args.cpu_tiles <= 0 else PARITY_TILES.get(cfg, 0),
                                             cpu_workers=args.cpu_workers))
                except Exception as exc:
                    lines.append({"config": cfg, "error": repr(exc)})
            out["configs"] = lines
            ints = []
            for kind in ("glass", "glass_mesh", "cast_ray"):
                try:
                    ints.append(integrator_line(kind, dev, parity_tiles=0 if args.no_parity or args.cpu_tiles <= 0 else 1024,
                                                cpu_workers=args.cpu_workers))
                except Exception as exc:
                    ints.append({"integrator": kind, "error": repr(exc)})
            out["integrators"] = ints
        if world == 1 and args.cpu_tiles > 0:
            sample = None
            try:
                if args.config == 4:
                    out["cpu_as_shipped"] = cpu_as_shipped()
                # the reference's compiled trace_path() on a bounded sample of THIS frame: the CPU baseline's timing, and
                # the pixels it renders are kept for the parity block below
                out["cpu_baseline"], sample = cpu_reference(args.config, W, H, spp, depth, sc.n_meshes, args.cpu_tiles,
                                                            args.cpu_workers, want_pixels=not args.no_parity)
            except Exception as exc:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "ray-bounces/s", "cores": 0, "kind": "reference",
                                       "sample": f"failed: {exc}"}
            if sample is not None and not args.no_parity and not args.shard:
                # north star: "matches the reference CPU render ... within 1e-4 per-channel RMS ... in the same run":
                # the frame the timed steps left in `image` against the reference's pixels (raytracer.c:197-221)
                try:
                    out["parity"] = gpu_parity(gs, sc, dev, sample, image, image8)
                except Exception as exc:
                    out["parity"] = {"ok": False, "error": repr(exc)}
        if world == 1:
            bad = [("headline", out["parity"])] if "parity" in out and not out["parity"].get("ok") else []
            bad += [(f"config {ln.get('config')}", ln["parity"]) for ln in out.get("configs", [])
                    if isinstance(ln.get("parity"), dict) and not ln["parity"].get("ok")]
            bad += [(ln.get("integrator"), ln["parity"]) for ln in out.get("integrators", [])
                    if isinstance(ln.get("parity"), dict) and not ln["parity"].get("ok")]
            if bad:
                out["error"] = "parity check failed: " + "; ".join(f"{w}: {json.dumps(p_)[:300]}" for w, p_ in bad)
                rc = 5

    # The single-process C path -- rt_hip_render_image, what render() of the raytracer.h boundary calls (reference
    # main.c:427-443): scene compare, launches, the frame over PCIe -- in a child of rank 0.  N > 1 on real GPUs: over the N
    # devices (grouped RCCL send / recv inside the shim) while every rank idles at the barrier below.  N = 1: the headline's
    # whole-call time (PCIe-inclusive; never `value`), then the same frame on 8 LOGICAL devices mapped onto this GPU -- the
    # multi-device path executed and compared bit for bit on the one GPU there is -- and the C command-line host's process
    # wall time by phase.  A failure or time-out is reported, never fatal.
    # (N = 1: with the other extras of the default line only -- `--no-configs` runs, i.e. profiling passes and A/Bs, skip them)
    if not rehearse and not args.shard and os.environ.get("RT_BENCH_HOST_PATH", "1") != "0" and (world > 1 or not args.no_configs):
        if rank == 0:
            out["host_path"] = run_host_path_child(args, world)
            if world == 1:
                out["host_path_logical8"] = run_host_path_child(args, 8, logical=True)
                out["cli_host"] = run_cli_host(args)
        if world > 1:
            dist.barrier(group=ctrl)
    if rank == 0:
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    gs.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
