#!/bin/bash
# Both bench lines (even and skewed index) without extras or CPU baseline: step and stage times (development aid, GPU box).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for d in uniform markov; do
  python3 bench.py --dist $d --no-extras --no-cpu-baseline > gpurun_out/two_bench_$d.json 2> gpurun_out/two_bench_$d.err
  python3 -c "
import json
d = json.load(open('gpurun_out/two_bench_$d.json'))
print('$d', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()})"
done
