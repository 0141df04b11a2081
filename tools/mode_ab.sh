#!/bin/bash
# The training step in one MMA mode with the product library against variant builds, on one box.
# Usage: tools/mode_ab.sh <tag> <mma mode> <variant .so> [...]
TAG=$1; MODE=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python bench.py --mma $MODE --no-secondary --no-cpu-baseline --steps 10 --warmup 3"
timeout -k 10 300 $B > $OUT/base_$MODE.json 2> $OUT/base_$MODE.err || exit 1
i=0
for lib in "$@"; do
  i=$((i+1))
  RSN_LIBRARY=$lib timeout -k 10 300 $B > $OUT/var${i}_$MODE.json 2> $OUT/var${i}_$MODE.err || { echo "variant $i failed"; tail -n 3 $OUT/var${i}_$MODE.err; }
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*_$MODE.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception:
        print(f, "no line"); continue
    k = j["train_step"]["kernels"]; lk = j["train_step"]["launch_kinds"]
    print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], " ".join("%s %.2f" % (n, v["ms_per_step"]) for n, v in k.items()),
          "other %.2f" % j["train_step"]["other_ms_per_step"], " ".join("%s %.3f" % (n.replace("field_", ""), v["avg_launch_ms"]) for n, v in lk.items()))
PY
