#!/bin/bash
# The round's evidence in ONE sequence on the GPU box, in the order the stamps need: the PMC passes first (they stamp
# profiles/scan_traffic.json with the SHA-256 of issl_kernels.hip), then the bench line that reads them.
#   tools/final_evidence.sh <tag>      -> gpurun_out/<tag>/{profiles/...,bench_n1.json,...}; copy what is wanted into profiles/
set -u
TAG=${1:-final}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG
mkdir -p "$OUT/profiles"
tools/profile_scan.sh "$TAG/prof" > "$OUT/prof.log" 2>&1; echo "prof rc=$?"
python3 tools/summarize_profile.py "$OUT/prof" profiles/r05_bench_n1 300000000 100000 uniform > "$OUT/prof_summary.txt" 2>&1; echo "summary rc=$?"
tools/profile_scan.sh "$TAG/prof10k" --guides 10000 > "$OUT/prof10k.log" 2>&1; echo "prof10k rc=$?"
python3 tools/summarize_profile.py "$OUT/prof10k" profiles/r05_scan_10k_guides_300m 300000000 10000 uniform > "$OUT/prof10k_summary.txt" 2>&1; echo "summary10k rc=$?"
python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; echo "bench rc=$?"
python3 bench.py --dist markov --no-extras > "$OUT/bench_n1_markov.json" 2> "$OUT/bench_n1_markov.err"; echo "markov rc=$?"
python3 bench.py --sites 50000000 --guides 10000 --no-extras > "$OUT/bench_10k_50m_uniform.json" 2>/dev/null; echo "50m rc=$?"
python3 bench.py --sites 50000000 --guides 10000 --dist markov --no-extras > "$OUT/bench_10k_50m_markov.json" 2>/dev/null; echo "50m markov rc=$?"
cp profiles/scan_traffic.json profiles/r05_bench_n1_pmc.json profiles/r05_bench_n1_kernel_stats.csv profiles/r05_scan_10k_guides_300m_pmc.json \
   profiles/r05_scan_10k_guides_300m_kernel_stats.csv "$OUT/profiles/"
ls "$OUT" "$OUT/profiles"
