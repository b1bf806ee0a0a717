#!/usr/bin/env python3
"""Collects the outputs of tools/gpu_round.sh (gpurun_out/) into profiles/<tag>_*:
bench JSON line, rocprofv3 --stats summary, the pt_* rows of the kernel trace, a PMC summary
and the traffic record bench.py reports.  usage: python tools/summarize_pmc.py r01f"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
vals = collections.OrderedDict()
for f in sorted(glob.glob(f"{G}/pmc_{tag}_*/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"] != "pt_render_tiles":
            continue
        vals.setdefault(row["Counter_Name"], []).append(
            (float(row["Counter_Value"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6))
bench = json.loads(open(f"{G}/bench_full.log").read().strip().splitlines()[-1])
casts = bench["ray_bounces_per_step"]
v = vals
ms = v["SQ_INSTS_VALU"][-1][1]
clk = v["GRBM_GUI_ACTIVE"][-1][0] / 8 / (v["GRBM_GUI_ACTIVE"][-1][1] * 1e-3)
busy = v["SQ_ACTIVE_INST_VALU"][-1][0] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"][-1][0] / 8)
util = v["SQ_THREAD_CYCLES_VALU"][-1][0] / (v["SQ_ACTIVE_INST_VALU"][-1][0] * 64)
fetch_kb, write_kb = v["FETCH_SIZE"][-1][0], v["WRITE_SIZE"][-1][0]
traffic = (2 * fetch_kb + write_kb) * 1024
cfg = bench["config"]
lines = [
    f"kernel pt_render_tiles, {cfg['workload']}, per launch",
    f"kernel time              {ms:.1f} ms",
    f"effective clock          {clk / 1e9:.3f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time)",
    f"VALU busy                {busy * 100:.1f} % (SQ_ACTIVE_INST_VALU [quad-cycles] * 4 / 1024 SIMDs / cycles; >= 100 % = saturated)",
    f"VALU lane utilisation    {util * 100:.1f} % (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64))",
    f"VALU wave-instructions   {v['SQ_INSTS_VALU'][-1][0]:.4g} = {v['SQ_INSTS_VALU'][-1][0] / casts * 64:.0f} per 64 ray-bounces",
    f"HBM read  (FETCH_SIZE)   {fetch_kb:.1f} KB; x2 (gfx950 correction for wide coalesced streams) = {2 * fetch_kb / 1024:.2f} MB",
    f"HBM write (WRITE_SIZE)   {write_kb / 1024:.2f} MB (algorithmic output: {15 * cfg['width'] * cfg['height'] / 1e6:.2f} MB)",
    f"HBM traffic              {traffic / 1e6:.1f} MB per launch -> {traffic / (ms * 1e-3) / 1e9:.3f} GB/s = {traffic / (ms * 1e-3) / 8e12:.2e} of 8 TB/s",
    "", "raw counters (value (kernel ms)), separate rocprofv3 --pmc passes:"]
lines += [f"{k:24s} " + "  ".join("%.5g (%.1f ms)" % x for x in vv) for k, vv in vals.items()]
open(f"{P}/{tag}_pmc_c4.txt", "w").write("\n".join(lines) + "\n")
json.dump({"kernel": "pt_render_tiles", "config": 4, "width": cfg["width"], "height": cfg["height"], "spp": cfg["spp"],
           "n_gpus": 1, "traffic_bytes_per_launch": traffic, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/gpu_round.sh); traffic = "
                     "(2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md",
           "source": f"profiles/{tag}_pmc_c4.txt"}, open(f"{P}/traffic_c4.json", "w"), indent=1)
json.dump(bench, open(f"{P}/{tag}_bench_c4.json", "w"))
st = glob.glob(f"{G}/prof_{tag}/*/*_kernel_stats.csv")[0]
shutil.copy(st, f"{P}/{tag}_bench_c4_kernel_stats.csv")
tr = glob.glob(f"{G}/prof_{tag}/*/*_kernel_trace.csv")[0]
rows = open(tr).read().splitlines()
open(f"{P}/{tag}_bench_c4_kernel_trace_pt.csv", "w").write("\n".join([rows[0]] + [r for r in rows[1:] if '"pt_' in r]) + "\n")
print("\n".join(lines[:10]))
