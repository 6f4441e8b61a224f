#!/bin/bash
# Same-box A/B of two builds on the small skewed workload (10 k guides x 50 M sites, --dist markov), where the chip is
# nearly empty behind the scan and a replay kernel lasts as long as its slowest guide: two alternating rounds.
cd "$(dirname "$0")/.."
cp crackling_amd/libissl_hip.so tools/_build/libissl_hip_cur.so
for round in 1 2; do
  for which in prev cur; do
    cp tools/_build/libissl_hip_$which.so crackling_amd/libissl_hip.so
    python3 bench.py --no-cpu-baseline --no-extras --dist markov --sites 50000000 --guides 10000 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('small markov $which', round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms'].items()})"
  done
done
cp tools/_build/libissl_hip_cur.so crackling_amd/libissl_hip.so
