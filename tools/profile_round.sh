#!/bin/bash
# Runs the rocprofv3 passes the roofline numbers come from (on the GPU box, via gpurun):
#   1. --kernel-trace --stats          per-kernel durations of the default bench.py run
#   2. --pmc FETCH_SIZE                HBM read side   (separate pass: TCC has 4 slots, FETCH_SIZE takes 3)
#   3. --pmc WRITE_SIZE                HBM write side
#   4. --pmc SQ_* (two passes)         MFMA busy cycles, clock, waits, instruction mix
# Usage: tools/profile_round.sh <tag> [extra bench.py arguments]
# Output: gpurun_out/prof_<tag>/...; summarise with tools/summarize_profile.py into profiles/.
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"   # extra bench.py arguments, e.g. --rays 16384 --samples 192 --mma bf16 (BASELINE configs[3])
OUT=/root/repo/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-secondary $EXTRA"
P="$B"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 20 --warmup 5 > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $P --steps 5 --warmup 2 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $P --steps 5 --warmup 2 > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -- $P --steps 5 --warmup 2 > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $P --steps 5 --warmup 2 > $OUT/pmc_sq2.log 2>&1 || exit 1
grep -h '"metric"' $OUT/stats.log > $OUT/bench_line.json
echo profile $TAG done
