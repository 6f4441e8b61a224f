#!/bin/bash
# rocprofv3 kernel tables of two builds (tools/_build/libissl_hip_prev.so against the one in the tree) on the small skewed
# workload (10 k guides x 50 M sites): which replay kernel a difference in `replay` comes from.   gpurun -- bash tools/kt_ab_small_markov.sh
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out/s12
cp crackling_amd/libissl_hip.so tools/_build/libissl_hip_cur.so
for which in prev cur; do
  cp tools/_build/libissl_hip_$which.so crackling_amd/libissl_hip.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s12/kt_$which -o kt -- python3 bench.py --no-cpu-baseline --no-extras --dist markov --sites 50000000 --guides 10000 > gpurun_out/s12/kt_$which.log 2>&1
  echo "== $which"; python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/s12/kt_$which/kt_kernel_stats.csv")))
for r in rows:
    if "replay" in r["Name"] or "prefix_single" in r["Name"] or "group_scatter" in r["Name"]: print(r["Name"][:50].ljust(50), r["Calls"].rjust(5), "%9.1f us"%(float(r["AverageNs"])/1e3))
PY
done
cp tools/_build/libissl_hip_cur.so crackling_amd/libissl_hip.so
