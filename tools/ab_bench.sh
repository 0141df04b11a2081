#!/bin/bash
# A/B on ONE box: the GPU parity tests, then the headline step with and without an environment switch.
# Usage: tools/ab_bench.sh <tag> <ENVVAR> [pytest -k expression]
TAG=$1; VAR=$2; KEXPR=${3:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "$KEXPR" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/steps.log
else
  timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/steps.log
fi
tail -5 $OUT/pytest.log
for rep in 1 2; do
  timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline > $OUT/bench_on_$rep.json 2> $OUT/bench_on_$rep.err; echo "on rc=$?" | tee -a $OUT/steps.log
  env $VAR=1 timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline > $OUT/bench_off_$rep.json 2> $OUT/bench_off_$rep.err; echo "off rc=$?" | tee -a $OUT/steps.log
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_o*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        k = j["train_step"]["kernels"]
        print(f.split("/")[-1], "ms/step %.2f" % j["ms_per_step"], " ".join("%s %.2f ms (%.3f)" % (n, v["ms_per_step"], v["frac_of_fp32_mfma_peak"]) for n, v in k.items()), "other %.2f" % j["train_step"]["other_ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
