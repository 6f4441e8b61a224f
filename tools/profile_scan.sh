#!/bin/bash
# rocprofv3 passes over the scoring pipeline (run on the GPU box through gpurun).
#   tools/profile_scan.sh <sites> <guides> <outdir-under-gpurun_out>
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2-4: PMC counters, each in its own run
# (never combined with tracing domains).  The program after `--` is python3 itself.
set -u
SITES=${1:-50000000}; GUIDES=${2:-10000}; OUT=gpurun_out/${3:-prof}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ARGS="tools/quick_perf.py --sites $SITES --guides $GUIDES --reps 5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $ARGS > "$OUT/kt.log" 2>&1
echo "kt rc=$?" >> "$OUT/kt.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
  --output-format csv -d "$OUT/pmc_sq" -o sq -- python3 $ARGS > "$OUT/pmc_sq.log" 2>&1
echo "sq rc=$?" >> "$OUT/pmc_sq.log"
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/pmc_sq2" -o sq2 -- python3 $ARGS > "$OUT/pmc_sq2.log" 2>&1
echo "sq2 rc=$?" >> "$OUT/pmc_sq2.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "fetch rc=$?" >> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
echo "write rc=$?" >> "$OUT/pmc_write.log"
ls -R "$OUT" | head -50
