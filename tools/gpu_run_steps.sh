#!/bin/bash
# Runs a list of GPU steps inside ONE gpurun call, each under its own `timeout -k 10`; a step that is killed at its limit
# (rc 124 / 137) ends the call -- no further GPU step after a timeout -- an ordinary failure does not.
# Usage: tools/gpu_run_steps.sh <tag> "<limit-seconds>|<name>|<command>" ...
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for spec in "$@"; do
  limit=${spec%%|*}; rest=${spec#*|}; name=${rest%%|*}; cmd=${rest#*|}
  echo "== $name: $cmd" >> $OUT/steps.log
  timeout -k 10 $limit bash -c "$cmd" > $OUT/$name.log 2> $OUT/$name.err
  rc=$?
  echo "$name rc=$rc" | tee -a $OUT/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at limit: stopping" | tee -a $OUT/steps.log; exit $rc; fi
done
exit 0
