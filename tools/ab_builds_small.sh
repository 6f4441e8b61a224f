#!/bin/bash
# Same-box A/B of two library builds at small batches (the HBM-bound regime of the scan): bench.py with 64 and 1000 guides
# per step against a 300 M-line index, alternating builds (see tools/ab_builds.sh for the large batches).
cp crackling_amd/libissl_hip.so tools/_build/libissl_hip_cur.so
for round in 1 2 3; do
  for which in prev cur; do
    cp tools/_build/libissl_hip_$which.so crackling_amd/libissl_hip.so
    for g in 64 1000; do
      python bench.py --guides $g --steps 200 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$g guides $which: step %.4f ms scan %.4f ms  %.0f GB/s physical' % (d['ms_per_step'], r['avg_launch_ms'], r['hbm_physical_GBps']))"
    done
  done
done
cp tools/_build/libissl_hip_cur.so crackling_amd/libissl_hip.so
