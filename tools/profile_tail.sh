#!/bin/bash
# rocprofv3 PMC passes aimed at the short kernels behind the scan (k_verify, k_group_scatter, k_replay, binning):
#   tools/profile_tail.sh <outdir-under-gpurun_out> [bench.py args...]
# L2 (TCC) request / hit / miss / atomic counts and SQ wave statistics, each set in its own run (never combined with tracing).
set -u
OUT=gpurun_out/${1:-prof_tail}; shift || true
ARGS="bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $*"
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
  --output-format csv -d "$OUT/sq" -o sq -- python3 $ARGS > "$OUT/sq.log" 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum \
  --output-format csv -d "$OUT/tcc" -o tcc -- python3 $ARGS > "$OUT/tcc.log" 2>&1
rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum \
  --output-format csv -d "$OUT/tcp" -o tcp -- python3 $ARGS > "$OUT/tcp.log" 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d "$OUT/mem" -o mem -- python3 $ARGS > "$OUT/mem.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $ARGS > "$OUT/kt.log" 2>&1
ls "$OUT"
