#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box differences on this pool reach 10 %, more than most changes):
#   build the baseline, cp crackling_amd/libissl_hip.so tools/_build/libissl_hip_prev.so, build the candidate, then
#   gpurun -- bash tools/ab_builds.sh   (bench.py at 10k guides and tools/quick_perf.py at 100k, three alternating rounds)
cp crackling_amd/libissl_hip.so tools/_build/libissl_hip_cur.so
for round in 1 2 3; do
  for which in prev cur; do
    cp tools/_build/libissl_hip_$which.so crackling_amd/libissl_hip.so
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('10k $which', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms'].items()})"
    python tools/quick_perf.py --sites 50000000 --guides 100000 --thr 75 --reps 5 2>&1 | grep -E "rep[34]" | cut -c1-90 | sed "s/^/100k $which /"
  done
done
cp tools/_build/libissl_hip_cur.so crackling_amd/libissl_hip.so
