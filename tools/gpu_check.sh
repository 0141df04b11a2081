#!/bin/bash
# One gpurun call: 2-rank DDP equivalence (before anything else touches the GPU), the GPU test suite, the default
# bench line and a 2-rank bench rehearsal on the shared GPU.  A step that is killed at its limit ends the call
# (no further GPU step after a timeout); an ordinary test failure does not.
# Usage: tools/gpu_check.sh [tag]
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() {  # run <limit-seconds> <name> <command...>
  local limit=$1 name=$2; shift 2
  echo "== $name" | tee -a $OUT/steps.log
  timeout -k 10 $limit "$@" > $OUT/$name.log 2> $OUT/$name.err
  local rc=$?
  echo "$name rc=$rc" | tee -a $OUT/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $OUT/steps.log; exit $rc; fi
  return $rc
}
export RSN_BENCH_SHARE_GPU=1
run 300 ddp_equiv python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/ddp_equiv.py --json $OUT/ddp_equiv.json
unset RSN_BENCH_SHARE_GPU
run 900 pytest_gpu python -m pytest tests -m gpu -q -s
run 300 bench_default python bench.py
export RSN_BENCH_SHARE_GPU=1
run 300 bench_n2_shared python bench.py --gpus 2 --steps 5 --warmup 2   # self-launching: no torchrun on the command line
unset RSN_BENCH_SHARE_GPU
tail -3 $OUT/pytest_gpu.log
tail -c 600 $OUT/bench_default.log
exit 0
