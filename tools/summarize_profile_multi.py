#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (tools/profile_round.sh) for SEVERAL kernels into profiles/<out>_summary.{md,json}:
per kernel the rocprofv3 average duration, the PMC counters (mean per launch) and the derived figures (HBM bytes corrected as
MI355X_MICROARCH.md prescribes, effective clock, MFMA busy, wave-wait fraction), plus optional algorithmic FLOP per launch ->
fraction of a peak.  Usage: summarize_profile_multi.py <tag> <out> "<kernel substring>[=<flop per launch>]" ..."""
import collections
import csv
import glob
import json
import os
import sys

tag, outname = sys.argv[1], sys.argv[2]
specs = [a.split("=") for a in sys.argv[3:]]
src = f"gpurun_out/prof_{tag}"
PEAK_TF = float(os.environ.get("RSN_PEAK_TFLOPS", "2500"))
f = max(glob.glob(f"{src}/stats/*/*_kernel_stats.csv"), key=os.path.getmtime)
stats = list(csv.DictReader(open(f)))
out = {"tag": tag, "kernel_stats": [{"name": r["Name"][:100], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                      "pct": float(r["Percentage"])} for r in stats], "kernels": {}}
for spec in specs:
    kern = spec[0]
    flop = float(spec[1]) if len(spec) > 1 else None
    fk = next((r for r in stats if kern in r["Name"]), None)
    if fk is None:
        continue
    o = {"avg_ms": float(fk["AverageNs"]) / 1e6, "calls": int(fk["Calls"])}
    pmc = {}
    for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        fs = glob.glob(f"{src}/{name}/*/*_counter_collection.csv")
        if not fs:
            continue
        agg, dur = collections.defaultdict(list), []
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in agg.items():
            pmc[k] = sum(v) / len(v)
        if dur:
            pmc[name + "_kernel_ms"] = sum(dur) / len(dur) / 1e6
    o["pmc"] = pmc
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        o["hbm_read_bytes_per_launch_corrected"] = pmc["FETCH_SIZE"] * 1024 * 2
        o["hbm_write_bytes_per_launch"] = pmc["WRITE_SIZE"] * 1024
        o["hbm_traffic_bytes_per_launch"] = o["hbm_read_bytes_per_launch_corrected"] + o["hbm_write_bytes_per_launch"]
        o["hbm_tbps"] = o["hbm_traffic_bytes_per_launch"] / (o["avg_ms"] * 1e-3) / 1e12
    if "GRBM_GUI_ACTIVE" in pmc:
        cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0
        o["effective_clock_ghz"] = cyc / (pmc["pmc_sq_kernel_ms"] * 1e-3) / 1e9
        o["mfma_busy_frac_of_simd_cycles"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
        o["wait_any_frac_of_wave_cycles"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
    if flop:
        o["algorithmic_flop_per_launch"] = flop
        o["achieved_tflops"] = flop / (o["avg_ms"] * 1e-3) / 1e12
        o["frac_of_peak"] = o["achieved_tflops"] / PEAK_TF
        o["peak_tflops"] = PEAK_TF
    out["kernels"][kern] = o
bl = f"{src}/bench_line.json"
if os.path.exists(bl):
    line = json.loads(open(bl).read().strip().splitlines()[-1])
    out["bench_line_under_rocprof"] = {k: line[k] for k in ("value", "unit", "ms_per_step", "dtype") if k in line}
    out["bench_kernels_under_rocprof"] = line.get("train_step", {}).get("kernels")
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(f"profiles/{outname}_summary.json", "w"), indent=1)
with open(f"profiles/{outname}_summary.md", "w") as w:
    w.write(f"# rocprofv3 summary {outname} (tools/profile_round.sh {tag}: python3 bench.py ...)\n\n")
    if "bench_line_under_rocprof" in out:
        w.write(f"bench line under rocprof: {json.dumps(out['bench_line_under_rocprof'])}\n\n")
    w.write("## --kernel-trace --stats\n\n| kernel | calls | avg us | % |\n|---|---|---|---|\n")
    for k in out["kernel_stats"][:16]:
        w.write(f"| `{k['name']}` | {k['calls']} | {k['avg_us']:.1f} | {k['pct']:.3f} |\n")
    for kern, o in out["kernels"].items():
        w.write(f"\n## `{kern}` (per launch, mean over {o['calls']} launches)\n\n")
        for k in ("avg_ms", "algorithmic_flop_per_launch", "achieved_tflops", "frac_of_peak", "hbm_read_bytes_per_launch_corrected",
                  "hbm_write_bytes_per_launch", "hbm_traffic_bytes_per_launch", "hbm_tbps", "effective_clock_ghz",
                  "mfma_busy_frac_of_simd_cycles", "wait_any_frac_of_wave_cycles"):
            if k in o:
                w.write(f"* {k}: {o[k]:.6g}\n")
        w.write("\n| counter | value |\n|---|---|\n")
        for k, v in sorted(o["pmc"].items()):
            w.write(f"| {k} | {v:.5g} |\n")
print(open(f"profiles/{outname}_summary.md").read()[:3000])
