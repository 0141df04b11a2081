#!/bin/bash
# timing diagnostics / A-B variants of the bf16 ring kernel (configs[3] size), built with -D macros (tools/_variant.py)
OUT=gpurun_out/ring_diag.log
: > $OUT
for d in ${DIAGS:-"" RSN_RING_NO_DMA RSN_RING_NO_BARRIER RSN_RING_NO_MFMA RSN_RING_NO_ENCODE RSN_RING_NO_WAIT}; do
  if [ -z "$d" ]; then timeout -k 10 300 python tools/variant_bench.py --mma bf16 --rays 16384 --samples 192 >> $OUT 2>&1 || exit 1
  else timeout -k 10 300 python tools/variant_bench.py --mma bf16 --rays 16384 --samples 192 $(for x in $d; do echo --define $x; done) >> $OUT 2>&1 || exit 1; fi
done
grep bf16 $OUT
