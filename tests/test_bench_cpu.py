"""CPU tests of bench.py's bookkeeping: the same-run parity comparison, the staleness rules that tie
committed PMC / PT_DIAG summaries to the kernel sources, the executed-work model, the tile samples."""
import copy
import json
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _sample(n=256, seed=1):
    rng = np.random.default_rng(seed)
    mean = rng.random((n, 3))
    rgb8 = (255.0 * np.clip(mean ** 0.2, 0, 1)).astype(np.uint8)
    return {"first": 3, "stride": 7, "count": n // 64, "px": np.arange(n, dtype=np.uint32), "rays": 1000, "casts": 900,
            "spp": 16, "oracle": "test", "seconds": 0.1, "mean": mean, "rgb8": rgb8}


def test_parity_numbers_green_and_each_way_it_can_fail():
    s = _sample()
    g = s["mean"].astype(np.float32)            # what the float framebuffer can hold
    ok = bench.parity_numbers(g, s["rgb8"], 1000, 900, s)
    assert ok["ok"] and ok["counters_equal"] and ok["pixels"] == 256 and max(ok["rms"]) < 1e-6 and ok["u8_max_diff"] == 0
    # one LSB in the bytes is inside the bar, two are not
    b1 = s["rgb8"].astype(np.int16)
    b1[5, 1] += 1 if b1[5, 1] < 255 else -1
    assert bench.parity_numbers(g, b1, 1000, 900, s)["ok"]
    b1[5, 1] += 2 if b1[5, 1] < 253 else -2
    assert not bench.parity_numbers(g, b1, 1000, 900, s)["ok"]
    # a value error beyond the RMS bar
    bad = g.copy()
    bad[:, 2] += 3e-4
    r = bench.parity_numbers(bad, s["rgb8"], 1000, 900, s)
    assert not r["ok"] and r["rms"][2] > bench.RMS_BAR >= r["rms"][0]
    # a decision that went the other way somewhere: counters differ, values may still agree
    r = bench.parity_numbers(g, s["rgb8"], 1001, 900, s)
    assert not r["ok"] and not r["counters_equal"]
    r = bench.parity_numbers(g, s["rgb8"], 1000, 899, s)
    assert not r["ok"] and not r["counters_equal"]
    # NaN anywhere is a failure, not a pass by comparison-is-false
    nan = g.copy()
    nan[0, 0] = np.nan
    assert not bench.parity_numbers(nan, s["rgb8"], 1000, 900, s)["ok"]


def test_sample_tiles_is_one_strided_run_inside_the_frame():
    for w, h, budget in ((1920, 1080, 2048), (800, 600, 1024), (256, 256, 1024), (3840, 2160, 2), (37, 21, 64), (8, 8, 5)):
        first, stride, count = bench.sample_tiles(w, h, budget)
        total = ((w + 7) // 8) * ((h + 7) // 8)
        assert count == min(budget, total) and stride >= 1 and first >= 0
        assert first + stride * (count - 1) < total
        px, slot, pit = bench.tile_pixel_indices(w, h, first, stride, count)
        assert len(np.unique(px)) == len(px) and px.max() < w * h
        assert slot.max() == count - 1 and pit.max() <= 63
        # the pixel a (slot, pixel-in-tile) pair names is the one the compact tile buffer holds there
        tx = (w + 7) // 8
        t = first + slot * stride
        assert (((t // tx) * 8 + pit // 8) * w + (t % tx) * 8 + pit % 8 == px).all()


def test_the_source_hash_covers_every_file_of_the_device_code():
    """the kernels are one translation unit (pt_kernel.hip) split by topic into pt_*.h headers: the hash that ties the
    committed PMC / PT_DIAG / ISA records to the code must see every file the two .hip sources include from the repo"""
    import re
    root = os.path.dirname(os.path.abspath(bench.__file__))
    csrc = os.path.join(root, "raytracer.c_amd", "csrc")
    hashed = {os.path.basename(f) for f in bench.KERNEL_SOURCES}
    seen, todo = set(), ["pt_kernel.hip", "rt_hip_shim.hip"]
    while todo:
        name = todo.pop()
        path = next((p_ for p_ in (os.path.join(csrc, name), os.path.join(root, "include", name)) if os.path.exists(p_)), None)
        if path is None or name in seen:
            continue            # a system header
        seen.add(name)
        todo += re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), re.M)
    assert seen <= hashed, sorted(seen - hashed)
    assert {f for f in os.listdir(csrc) if re.fullmatch(r"pt_\w+\.h", f)} <= hashed


def test_committed_pmc_records_are_tied_to_the_kernel_sources():
    sha = bench.kernel_source_sha256()
    assert len(sha) == 64 and sha == bench.kernel_source_sha256()
    rec = {"file": "profiles/pmc_cX.json", "spp": 1024, "kernel_ms": 200.0, "source_sha256": sha,
           "valu_busy": 1.0, "lane_utilisation": 0.67, "valu_instr_per_64_bounces": 850.0, "traffic_bytes_per_launch": 3.5e7}
    # fresh: same sources, kernel time within 3 %
    assert bench.pmc_is_stale(rec, 204.0, 1024) is None
    keys = bench.pmc_keys(rec, 1024, 204.0)
    assert keys["pmc_stale"] is False and keys["valu_busy"] == 1.0 and keys["traffic"] == 3.5e7
    # stale by time: the kernel got faster (or slower) by more than 3 % since the pass
    assert bench.pmc_is_stale(rec, 206.1, 1024) is not None and bench.pmc_is_stale(rec, 193.9, 1024) is not None
    assert "kernel time" in bench.pmc_is_stale(rec, 190.0, 1024)
    # sub-millisecond kernels get 20 microseconds of clock-ramp noise on top of the 3 %
    short = dict(rec, kernel_ms=0.872, spp=64)
    assert bench.pmc_is_stale(short, 0.895, 64) is None and bench.pmc_is_stale(short, 0.93, 64) is not None
    keys = bench.pmc_keys(rec, 1024, 190.0)
    assert keys["pmc_stale"] is True and keys["valu_busy"] is None and keys["traffic"] is None and "kernel time" in keys["pmc_stale_reason"]
    # stale by source: any edit of the kernel sources
    old = dict(rec, source_sha256="0" * 64)
    assert "source hash" in bench.pmc_is_stale(old, 200.0, 1024)
    assert bench.pmc_keys(old, 1024, 200.0)["lane_utilisation"] is None
    # a record from before the hashes existed is stale too
    legacy = {k: v for k, v in rec.items() if k not in ("source_sha256", "kernel_ms")}
    assert bench.pmc_is_stale(legacy, 200.0, 1024) is not None
    # a pass at another spp: per-sample time within 8 %, traffic scaled
    assert bench.pmc_is_stale(rec, 200.0 * 4 * 1.05, 4096) is None
    assert bench.pmc_is_stale(rec, 200.0 * 4 * 1.12, 4096) is not None
    assert bench.pmc_keys(rec, 4096, 200.0 * 4)["traffic"] == 4 * 3.5e7
    # no record at all: nulls, and not "stale"
    assert bench.pmc_keys(None, 1024, 200.0)["valu_busy"] is None and "pmc_stale" not in bench.pmc_keys(None, 1024, 200.0)


def test_the_committed_summaries_parse_and_say_what_they_were_measured_on():
    for cfg in (2, 3, 4, 5):
        rec = bench.committed_pmc(cfg, *{2: (800, 600), 3: (1920, 1080), 4: (1920, 1080), 5: (3840, 2160)}[cfg], 0, 1, any_spp=True)
        assert rec is not None and rec["kernel"].startswith("pt_render_tiles")
        # whichever way the check comes out today, it must come out as a reason or None -- never raise
        r = bench.pmc_is_stale(rec, rec.get("kernel_ms", 1.0), rec["spp"])
        assert r is None or isinstance(r, str)


def test_executed_work_model():
    sha = bench.kernel_source_sha256()
    per = {k: 0.0 for k in bench.EXEC_FLOPS}
    per.update({"exact_sphere": 2.0, "filter_sphere": 30.0, "hit": 1.0})
    diag = {"file": "profiles/diag_cX.json", "source_sha256": sha, "width": 8, "height": 8, "spp": 1, "kernel": "k",
            "per_ray_bounce": per}
    out = bench.executed_work(diag, casts=1e10, kernel_s=0.25, n_spheres=38, n_triangles=0)
    f64 = 2 * 17.0 + 29.0
    f32 = 30 * 15.0
    assert out["executed_flops_per_ray_bounce"] == {"f64": f64, "f32": f32}
    want = 4e10 * (f64 / 39.3e12 + f32 / 157.3e12)
    assert abs(out["frac_executed"] - want) < 1e-12 and "frac_hierarchy_model" not in out
    # hierarchy scenes also get the algorithmic model with the walk in place of the O(N) triangle scan
    per2 = dict(per, node_visit=0.5, leaf_pretest=0.3, exact_triangle=0.01, probe=0.8)
    out2 = bench.executed_work(dict(diag, per_ray_bounce=per2), 1e10, 0.25, 8, 10240)
    a64 = 17.0 * 8 + 120.0 + 0.01 * 40
    a32 = 0.5 * 36 + 0.3 * 51 + 0.8 * 13
    assert abs(out2["frac_hierarchy_model"] - 4e10 * (a64 / 39.3e12 + a32 / 157.3e12)) < 1e-12
    # counted on other sources: no number, and it says why
    stale = bench.executed_work(dict(diag, source_sha256="f" * 64), 1e10, 0.25, 38, 0)
    assert stale["frac_executed"] is None and stale["diag_stale"] is True and "stale" in stale["executed_note"]
    assert bench.executed_work(None, 1e10, 0.25, 38, 0)["frac_executed"] is None


def test_isa_and_diag_records_follow_the_sources_too():
    """the committed ISA statistics and PT_DIAG counts are reported only while they describe today's kernel sources"""
    sha = bench.kernel_source_sha256()
    rec = json.load(open(os.path.join(ROOT, "profiles", "isa_stats.json")))
    assert set(rec["kernels"]) >= {"pt_render_tiles", "pt_render_tiles_tri", "pt_render_tiles_tri_queued_sph", "pt_render_tiles_refr_pool",
                                   "pt_render_tiles_pool_mem_s", "pt_whitted_tiles"}
    k = bench.isa_keys("pt_render_tiles", source_sha=rec["source_sha256"])
    assert k["vgpr"] > 0 and k["scratch_bytes"] == 0 and "isa_stale" not in k
    assert bench.isa_keys("pt_render_tiles", source_sha="0" * 64) == {"isa_stale": True}
    assert bench.isa_keys("no_such_kernel", source_sha=rec["source_sha256"]) is None
    # no shipped kernel uses scratch memory: whatever the hash says today, the committed listing must say so
    assert all(v["scratch_bytes"] == 0 and v["scratch_ops"] == 0 for v in rec["kernels"].values()), \
        {n: v["scratch_bytes"] for n, v in rec["kernels"].items() if v["scratch_bytes"]}
    for cfg in (1, 2, 3, 4, 5):
        d = bench.committed_diag(cfg)
        assert d is not None and set(d["per_ray_bounce"]) == set(bench.EXEC_FLOPS)
        fresh = bench.executed_work(d, 1e9, 0.1, 8, 0, source_sha=d["source_sha256"])
        assert 0 < fresh["frac_executed"] < 1
    assert sha  # (whether the committed records are fresh TODAY is the profile session's business; bench.py says so in its line)
