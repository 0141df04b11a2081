#!/bin/bash
# A/B of ring-kernel build variants with bench.py's event-timed kernel duration (configs[3]); each arm twice, interleaved
V=$(python -c "import sys; sys.path.insert(0,'.'); from tools._variant import build_variant; print(build_variant('$1'.split()))") || exit 1
for i in 1 2; do
  for lib in reflect_sampling_nerf_amd/librsn_hip.so $V; do
    RSN_LIBRARY=$lib timeout -k 10 120 python bench.py --workload level --rays 16384 --samples 192 --mma bf16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'[-20:], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))" || exit 1
  done
done
