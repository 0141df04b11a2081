"""One training step of a rocprofv3 kernel trace (tools/profile_round.sh <tag>): span, busy and idle time, the field /
weight-gradient launches in order, and the small launches grouped by kernel.
    python tools/step_timeline.py gpurun_out/prof_r03_train > profiles/r03_step_timeline.txt"""
import collections, csv, glob, os, sys

src = sys.argv[1]
f = max(glob.glob(src + "/stats/*/*_kernel_trace.csv"), key=os.path.getmtime)  # newest run (gpurun merges into old directories)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rsn_radam" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]  # the last complete step: behind one RAdam launch up to and including the next
t0 = int(step[0]["Start_Timestamp"])
busy = idle = 0
prev = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    idle += max(0, s - prev)
    busy += e - s
    prev = max(prev, e)
print("# one training step of %s (under rocprofv3: short launches read ~5 %% longer than by HIP events)" % f.split("/")[-4])
print("launches %d   span %.3f ms   kernel time %.3f ms   idle between kernels %.3f ms" % (len(step), (prev - t0) / 1e6, busy / 1e6, idle / 1e6))
print("\n# field / weight-gradient launches, in order")
big = ("rsn_field", "rsn_wgrad")
for r in step:
    n = r["Kernel_Name"]
    if any(b in n for b in big):
        print("%-66s %9.1f us   grid %s" % (n[:66], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]))
small = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = r["Kernel_Name"]
    if not any(b in n for b in big):
        small[n[:70]][0] += 1
        small[n[:70]][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("\n# everything else: %d launches, %.1f us" % (sum(v[0] for v in small.values()), sum(v[1] for v in small.values())))
for k, v in sorted(small.items(), key=lambda kv: -kv[1][1]):
    print("  %-70s x%-3d %7.1f us" % (k, v[0], v[1]))
