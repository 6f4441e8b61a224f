#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun):
#   tools/profile_scan.sh <outdir-under-gpurun_out> [bench.py args...]
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2-5: PMC counters, each in its own run
# (never combined with tracing domains; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# The program after `--` is python3 itself (no env/bash wrappers under the profiler).  bench.py runs without the CPU
# baseline and without its extra measurement points, so that every k_scan launch of a pass has the same batch size.
set -u
OUT=gpurun_out/${1:-prof}; shift || true
ARGS="bench.py --no-cpu-baseline --no-extras $*"
cd "$(dirname "$0")/.."
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $ARGS > "$OUT/kt.log" 2>&1
echo "kt rc=$?" >> "$OUT/kt.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
  --output-format csv -d "$OUT/pmc_sq" -o sq -- python3 $ARGS > "$OUT/pmc_sq.log" 2>&1
echo "sq rc=$?" >> "$OUT/pmc_sq.log"
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/pmc_sq2" -o sq2 -- python3 $ARGS > "$OUT/pmc_sq2.log" 2>&1
echo "sq2 rc=$?" >> "$OUT/pmc_sq2.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "fetch rc=$?" >> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
echo "write rc=$?" >> "$OUT/pmc_write.log"
ls "$OUT"
