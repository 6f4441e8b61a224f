#!/bin/bash
# rocprofv3 PMC passes aimed at the short kernels behind the scan (k_verify, k_group_scatter, k_replay, binning):
#   tools/profile_tail.sh <outdir-under-gpurun_out> [bench.py args...]
# L2 (TCC) request / hit / miss counts and SQ wave statistics, each small set in its own run (never combined with
# tracing), every run under its own timeout, a progress line after each.
set -u
OUT=gpurun_out/${1:-prof_tail}; shift || true
ARGS="bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $*"
cd "$(dirname "$0")/.."
mkdir -p "$OUT"
export TMPDIR=/tmp
pass() { # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass tcc2 TCC_ATOMIC_sum TCC_READ_sum TCC_WRITE_sum
pass tcp1 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
pass mem FETCH_SIZE
pass memw WRITE_SIZE
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $ARGS > "$OUT/kt.log" 2>&1
echo "pass kt rc=$?"
