#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box differences on this pool reach 10 %, more than most changes):
#   build the baseline, cp crackling_amd/libissl_hip.so tools/_build/libissl_hip_prev.so, build the candidate, then
#   gpurun -- bash tools/ab_builds.sh   (tools/quick_perf.py at 10k x 50M and 100k x 300M, three alternating rounds)
cp crackling_amd/libissl_hip.so tools/_build/libissl_hip_cur.so
for round in 1 2 3; do
  for which in prev cur; do
    cp tools/_build/libissl_hip_$which.so crackling_amd/libissl_hip.so
    python tools/quick_perf.py --sites 50000000 --guides 10000 --thr 75 --reps 6 2>&1 | grep -E "rep[3-5]" | cut -c1-110 | sed "s/^/10k  $which /"
    python tools/quick_perf.py --fast-synth --sites 300000000 --guides 100000 --thr 75 --reps 4 2>&1 | grep -E "rep[23]" | cut -c1-110 | sed "s/^/100k $which /"
  done
done
cp tools/_build/libissl_hip_cur.so crackling_amd/libissl_hip.so
