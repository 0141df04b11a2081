#!/bin/bash
# PMC counters of one weight-gradient shape:  bash tools/wgrad_pmc.sh <mode> <out-dir-under-gpurun_out>
MODE=${1:-bf16x6}; OUT=gpurun_out/${2:-wgrad_pmc}
mkdir -p $OUT; export TMPDIR=/tmp
cat > $OUT/run.py <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import train_graph
train_graph._WGRAD_MODE = {"f32": 0, "bf16x6": 1, "bf16": 3}["$MODE"]
pkg.load_library()
dev = torch.device("cuda", 0)
n = 524288
dy = torch.randn(n, 256, device=dev); x = torch.randn(n, 256, device=dev)
dw = torch.zeros(256, 256, device=dev); db = torch.zeros(256, device=dev)
for _ in range(6):
    train_graph._wgrad(dy, 256, x, 256, dw, 0, db)
torch.cuda.synchronize()
PY
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq -- python3 $OUT/run.py > $OUT/sq.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $OUT/run.py > $OUT/sq2.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/trace.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for sub in ("sq", "sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "wgrad" not in r["Kernel_Name"]: continue
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(r["Kernel_Name"][:60], r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(k)
        for c, v in d.items(): print("   %-28s %.4g per launch" % (c, v / cnt[(k, c)]))
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
