#!/bin/bash
# clock and MFMA-busy of the staged x6 kernel, product vs ablations
# Usage (on the GPU box): bash tools/wgrad_clock.sh [variant .so built HERE with tools/_variant.py ...]
OUT=gpurun_out/wgclk; mkdir -p $OUT; export TMPDIR=/tmp
cat > $OUT/run.py <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import reflect_sampling_nerf_amd as pkg
from reflect_sampling_nerf_amd import train_graph
train_graph._WGRAD_MODE = 1
lib = os.environ.get("WG_LIB")
pkg.load_library(lib) if lib else pkg.load_library()
dev = torch.device("cuda", 0)
n = 524288
dy = torch.randn(n, 256, device=dev); x = torch.randn(n, 256, device=dev)
dw = torch.zeros(256, 256, device=dev); db = torch.zeros(256, device=dev)
for _ in range(30):
    train_graph._wgrad(dy, 256, x, 256, dw, 0, db)
torch.cuda.synchronize()
PY
i=0
for lib in "" "$@"; do
  export WG_LIB=$lib
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq$i -- python3 $OUT/run.py > $OUT/sq$i.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter(); dur=[]
for f in glob.glob("$OUT/sq$i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for f in glob.glob("$OUT/sq$i/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
d = sorted(dur)[len(dur)//2]
m = {k: v/cnt[k] for k, v in agg.items()}
print("lib=$lib", "median us %.1f" % d, "clock GHz %.3f" % (m["GRBM_GUI_ACTIVE"]/8/d/1e3), "mfma busy %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"]/ (m["GRBM_GUI_ACTIVE"]/8) / (256*4) if "SQ_VALU_MFMA_BUSY_CYCLES" in m else -1), {k: "%.4g"%v for k,v in m.items()})
PY
  i=$((i+1))
done
