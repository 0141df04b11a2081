#!/bin/bash
# timing diagnostics of the training kernels (headline step): builds variants of the library with -D macros
# (tools/_variant.py) and prints the per-kernel times bench.py measures.  Wrong results by construction.
OUT=gpurun_out/train_diag.log
: > $OUT
for d in ${DIAGS:-"" RSN_DIAG_NO_SAVED_ROWS RSN_DIAG_WG_NO_FLUSH}; do
  if [ -z "$d" ]; then LIB=""; else LIB=$(python -c "import sys; sys.path.insert(0,'.'); from tools._variant import build_variant; print(build_variant(['$d']))") || exit 1; fi
  echo "variant [$d]" >> $OUT
  RSN_LIBRARY=${LIB:-reflect_sampling_nerf_amd/librsn_hip.so} timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --steps 8 --warmup 2 2>>$OUT | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'], 2), {k: round(v['ms_per_step'], 2) for k, v in d['train_step']['kernels'].items()}, {k: round(v['avg_launch_ms'], 3) for k, v in d['train_step']['launch_kinds'].items()})" >> $OUT || exit 1
done
cat $OUT
