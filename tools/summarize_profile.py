#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (tools/profile_round.sh) into profiles/<tag>_summary.{md,json}."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "rsn_field_kernel<8, false, 0>"  # exact-fp32 eval instantiation
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
out = {"tag": tag, "kernel": KERNEL}
try:  # workload size of the profiled bench run
    bl = json.loads(open(f"gpurun_out/prof_{tag}/bench_line.json").readline())
    out["rays"], out["samples"] = bl["config"]["rays_per_gpu"], bl["config"]["samples_per_ray"]
    out["bench_line"] = {k: bl[k] for k in ("value", "unit", "ms_per_step", "dtype") if k in bl}
except (OSError, ValueError, KeyError):
    pass
# 1. kernel stats
f = max(glob.glob(f"{src}/stats/*/*_kernel_stats.csv"), key=os.path.getmtime)  # newest run (gpurun merges into old dirs)
stats = list(csv.DictReader(open(f)))
out["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                        "pct": float(r["Percentage"])} for r in stats]
fk = next(r for r in stats if KERNEL in r["Name"])
out["field_kernel_avg_ms"] = float(fk["AverageNs"]) / 1e6
# 2. PMC
pmc = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    fs = glob.glob(f"{src}/{name}/*/*_counter_collection.csv")
    if not fs:
        continue
    agg, dur = collections.defaultdict(list), []
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
    pmc[name + "_kernel_ms"] = sum(dur) / len(dur) / 1e6
out["pmc_field_kernel"] = pmc
# HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE/WRITE_SIZE are in KiB-ish
# units of 1024 B; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced streaming read -> doubled (upper bound:
# this kernel's reads are narrow ray/bin loads plus L2-resident weights).
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    out["hbm_read_bytes_per_launch_corrected"] = pmc["FETCH_SIZE"] * 1024 * 2
    out["hbm_write_bytes_per_launch"] = pmc["WRITE_SIZE"] * 1024
    out["hbm_traffic_bytes_per_launch"] = out["hbm_read_bytes_per_launch_corrected"] + out["hbm_write_bytes_per_launch"]
if "GRBM_GUI_ACTIVE" in pmc:
    cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0
    out["effective_clock_ghz"] = cyc / (pmc["pmc_sq_kernel_ms"] * 1e-3) / 1e9
    out["mfma_busy_frac_of_simd_cycles"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
    out["wait_any_frac_of_wave_cycles"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
bl = f"{src}/bench_line.json"
if os.path.exists(bl):
    out["bench_line_under_rocprof"] = json.loads(open(bl).read().strip().splitlines()[-1])
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
with open(f"profiles/{tag}_summary.md", "w") as w:
    w.write(f"# rocprofv3 summary {tag} (python3 bench.py, default workload)\n\n")
    w.write("## --kernel-trace --stats\n\n| kernel | calls | avg us | % |\n|---|---|---|---|\n")
    for k in out["kernel_stats"]:
        w.write(f"| `{k['name']}` | {k['calls']} | {k['avg_us']:.1f} | {k['pct']:.3f} |\n")
    w.write(f"\n## PMC, {KERNEL} (per launch, mean)\n\n| counter | value |\n|---|---|\n")
    for k, v in sorted(pmc.items()):
        w.write(f"| {k} | {v:.5g} |\n")
    w.write("\n## derived\n\n")
    for k in ("field_kernel_avg_ms", "hbm_read_bytes_per_launch_corrected", "hbm_write_bytes_per_launch",
              "hbm_traffic_bytes_per_launch", "effective_clock_ghz", "mfma_busy_frac_of_simd_cycles",
              "wait_any_frac_of_wave_cycles"):
        if k in out:
            w.write(f"* {k}: {out[k]:.6g}\n")
print(open(f"profiles/{tag}_summary.md").read())
