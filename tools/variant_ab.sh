#!/bin/bash
# Headline step with the product library against variant builds (extra -D macros; built HERE beforehand with
# `python -c "from tools._variant import build_variant; print(build_variant([...]))"`), on one box, interleaved.
# Usage: tools/variant_ab.sh <tag> <variant .so> [<variant .so> ...]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3"
for rep in 1 2; do
  timeout -k 10 300 $B > $OUT/base_$rep.json 2> $OUT/base_$rep.err || exit 1
  i=0
  for lib in "$@"; do
    i=$((i+1))
    RSN_LIBRARY=$lib timeout -k 10 300 $B > $OUT/var${i}_$rep.json 2> $OUT/var${i}_$rep.err || exit 1
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1]); k = j["train_step"]["kernels"]
    print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], " ".join("%s %.2f (%.3f)" % (n, v["ms_per_step"], v["frac_of_fp32_mfma_peak"]) for n, v in k.items()), "other %.2f" % j["train_step"]["other_ms_per_step"], j["train_step"]["launch_kinds"]["field_forward_train_normals"]["avg_launch_ms"])
PY
