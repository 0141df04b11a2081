#!/bin/bash
# A/B of the bf16 ring kernel on BASELINE configs[3] (16384 x 192): start skew between the CUs of an XCD
OUT=gpurun_out/ring_sweep.log
: > $OUT
for st in ${STAGGERS:-0 1 2 4}; do
  echo "stagger $st" >> $OUT
  RSN_RING_STAGGER=$st timeout -k 10 120 python bench.py --workload level --rays 16384 --samples 192 --mma bf16 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['roofline']['frac'])" >> $OUT || exit 1
done
cat $OUT
