#!/bin/bash
# The plain-bf16 training step with the product library against variant builds (extra -D macros; built HERE beforehand with
# `python -c "from tools._variant import build_variant; print(build_variant([...]))"`), on one box.
# Usage: [RSN_AB_MMA=bf16x6] tools/bf16_train_ab.sh <tag> <variant .so> [<variant .so> ...]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python bench.py --mma ${RSN_AB_MMA:-bf16} --no-secondary --no-cpu-baseline --steps 10 --warmup 3"
timeout -k 10 300 $B > $OUT/base.json 2> $OUT/base.err || exit 1
i=0
for lib in "$@"; do
  i=$((i+1))
  RSN_LIBRARY=$lib timeout -k 10 300 $B > $OUT/var$i.json 2> $OUT/var$i.err || { echo "variant $i ($lib) failed"; tail -n 3 $OUT/var$i.err; }
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception:
        print(f, "no line"); continue
    k = j["train_step"]["kernels"]; lk = j["train_step"]["launch_kinds"]
    print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], " ".join("%s %.2f" % (n, v["ms_per_step"]) for n, v in k.items()),
          "other %.2f" % j["train_step"]["other_ms_per_step"], " ".join("%s %.3f" % (n.replace("field_", ""), v["avg_launch_ms"]) for n, v in lk.items()))
PY
