#!/usr/bin/env python3
"""Collects tools/gpu_pmc_cfg.sh's outputs (gpurun_out/cfg<C>_<tag>/) into profiles/<tag>_c<C>_*:
the bench JSON line, rocprofv3 --stats summary and a PMC summary of the dominant render kernel.
usage: python tools/summarize_pmc_cfg.py r02a 5"""
import collections, csv, glob, json, os, shutil, sys
tag, cfg_n = sys.argv[1], int(sys.argv[2])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha256  # ties the record to the kernel sources it was measured on (bench.py: pmc_is_stale)
D, P = os.path.join(ROOT, "gpurun_out", f"cfg{cfg_n}_{tag}"), os.path.join(ROOT, "profiles")
bench = json.loads(open(f"{D}/bench.log").read().strip().splitlines()[-1])
kernel = bench["roofline"]["kernel"]
vals = collections.OrderedDict()
for f in sorted(glob.glob(f"{D}/pmc_*/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"] != kernel:
            continue
        vals.setdefault(row["Counter_Name"], []).append(
            (float(row["Counter_Value"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6))
casts = bench["ray_bounces_per_step"]
v = vals
ms = v["SQ_INSTS_VALU"][-1][1]
clk = v["GRBM_GUI_ACTIVE"][-1][0] / 8 / (v["GRBM_GUI_ACTIVE"][-1][1] * 1e-3)
# VALU busy: SQ_ACTIVE_INST_VALU counts quad-cycles summed over WAVES (a SIMD that overlaps two waves' VALU instructions --
# packed-fp32 / 32-bit ops issue in 2 of the 4 cycles -- counts both), normalised by the cycles of the SAME pass
# (GRBM_GUI_ACTIVE is collected with the SQ counters).  So the raw ratio can exceed 1: it is reported as measured, and
# `valu_busy` is that ratio capped at 1 ("the VALU issue port is saturated"), not a 3-digit measurement above 100 %.
busy_raw = v["SQ_ACTIVE_INST_VALU"][-1][0] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"][-1][0] / 8)
busy = min(busy_raw, 1.0)
sq_busy = v["SQ_BUSY_CYCLES"][-1][0] / v["GRBM_GUI_ACTIVE"][-1][0] if "SQ_BUSY_CYCLES" in v else None
util = v["SQ_THREAD_CYCLES_VALU"][-1][0] / (v["SQ_ACTIVE_INST_VALU"][-1][0] * 64)
fetch_kb, write_kb = v["FETCH_SIZE"][-1][0], v["WRITE_SIZE"][-1][0]
traffic = (2 * fetch_kb + write_kb) * 1024
cfg = bench["config"]
lines = [
    f"kernel {kernel}, {cfg['workload']}, per launch",
    f"kernel time              {ms:.2f} ms",
    f"effective clock          {clk / 1e9:.3f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time)",
    f"VALU busy                {busy * 100:.1f} % (raw ratio {busy_raw * 100:.1f} %: SQ_ACTIVE_INST_VALU [quad-cycles, summed over waves] * 4 / 1024 SIMDs / "
    f"GRBM_GUI_ACTIVE per XCD of the same pass; a raw value above 100 % = overlapping waves counted twice, i.e. saturated)",
    f"VALU lane utilisation    {util * 100:.1f} % (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64))",
    f"VALU wave-instructions   {v['SQ_INSTS_VALU'][-1][0]:.4g} = {v['SQ_INSTS_VALU'][-1][0] / casts * 64:.0f} per 64 ray-bounces",
    f"LDS wave-instructions    {v['SQ_INSTS_LDS'][-1][0]:.4g} = {v['SQ_INSTS_LDS'][-1][0] / casts * 64:.0f} per 64 ray-bounces",
    f"SALU wave-instructions   {v['SQ_INSTS_SALU'][-1][0]:.4g} = {v['SQ_INSTS_SALU'][-1][0] / casts * 64:.0f} per 64 ray-bounces",
    f"HBM read  (FETCH_SIZE)   {fetch_kb:.1f} KB; x2 (gfx950 correction for wide coalesced streams) = {2 * fetch_kb / 1024:.2f} MB",
    f"HBM write (WRITE_SIZE)   {write_kb / 1024:.2f} MB (algorithmic output: {15 * cfg['width'] * cfg['height'] / 1e6:.2f} MB)",
    f"HBM traffic              {traffic / 1e6:.1f} MB per launch -> {traffic / (ms * 1e-3) / 1e9:.3f} GB/s = {traffic / (ms * 1e-3) / 8e12:.2e} of 8 TB/s",
    "", "raw counters (value (kernel ms)), separate rocprofv3 --pmc passes:"]
lines += [f"{k:24s} " + "  ".join("%.5g (%.2f ms)" % x for x in vv) for k, vv in vals.items()]
pre = f"{P}/{tag}_c{cfg_n}"
open(f"{pre}_pmc.txt", "w").write("\n".join(lines) + "\n")
json.dump({"kernel": kernel, "config": cfg_n, "width": cfg["width"], "height": cfg["height"], "spp": cfg["spp"],
           "n_gpus": 1, "source_sha256": kernel_source_sha256(), "kernel_ms": bench["roofline"]["kernel_ms"],
           "kernel_ms_note": "HIP-event kernel time of the un-profiled bench run of the same session (gpurun_out/cfg<C>_<tag>/bench.log); "
                             "bench.py reports these counters only while the sources hash to source_sha256 and its own kernel "
                             "time agrees within 3 %",
           "pmc_pass_kernel_ms": ms, "traffic_bytes_per_launch": traffic, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
           "valu_busy": busy, "valu_busy_raw": busy_raw,
           "valu_busy_note": "SQ_ACTIVE_INST_VALU*4/1024/(GRBM_GUI_ACTIVE/8), both from ONE pass; per-wave quad-cycles, so overlapping "
                             "waves on a SIMD can push the raw ratio past 1: valu_busy is capped at 1 = saturated",
           "lane_utilisation": util, "valu_instr_per_64_bounces": v["SQ_INSTS_VALU"][-1][0] / casts * 64,
           "effective_clock_ghz": clk / 1e9,
           "method": "rocprofv3 --pmc, one counter group per pass with --kernel-trace only (tools/gpu_pmc_cfg.sh); traffic = "
                     "(2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md",
           "source": f"profiles/{tag}_c{cfg_n}_pmc.txt"}, open(f"{P}/pmc_c{cfg_n}.json", "w"), indent=1)
json.dump(bench, open(f"{pre}_bench.json", "w"))
st = glob.glob(f"{D}/prof/*/*_kernel_stats.csv")[0]
shutil.copy(st, f"{pre}_kernel_stats.csv")
print("\n".join(lines[:12]))
