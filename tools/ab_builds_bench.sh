#!/bin/bash
# Same-box A/B of two builds of the library through bench.py itself (see tools/ab_builds.sh for the recipe), even and skewed index:
#   gpurun -- bash tools/ab_builds_bench.sh   (bench.py without extras, two alternating rounds each)
cd "$(dirname "$0")/.."
cp crackling_amd/libissl_hip.so tools/_build/libissl_hip_cur.so
for dist in uniform markov; do
  for round in 1 2; do
    for which in prev cur; do
      cp tools/_build/libissl_hip_$which.so crackling_amd/libissl_hip.so
      python3 bench.py --no-cpu-baseline --no-extras --dist $dist 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$dist $which', round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms'].items()})"
    done
  done
done
cp tools/_build/libissl_hip_cur.so crackling_amd/libissl_hip.so
