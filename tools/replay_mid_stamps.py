#!/usr/bin/env python3
"""Digest of k_replay_mid's phase clocks in an ISSL_SCAN_STAMPS dump (16 u64 per listed guide behind the scan's records;
entries with [14] == 1 are k_replay_mid's):
    ISSL_SCAN_STAMPS=/tmp/st.bin python bench.py --dist markov --no-extras --no-cpu-baseline --steps 3 --warmup 1
    python tools/replay_mid_stamps.py /tmp/st.bin"""
import numpy as np, sys
a = np.fromfile(sys.argv[1], dtype=np.uint64)[524288:524288 + 16 * 4096].reshape(-1, 16).astype(np.int64)
a = a[(a[:, 0] > 0) & (a[:, 14] == 1) & (a[:, 7] > 0)]
us = lambda x: x / 100.0
print("mid guides stamped", len(a), "hits p50/p90", np.percentile(a[:, 1], [50, 90]), "first slice p50/p90", np.percentile(a[:, 4], [50, 90]),
      "scored p50/p90", np.percentile(a[:, 8], [50, 90]), "walked (hits of the slices touched) p50/p90", np.percentile(a[:, 9], [50, 90]))
for name, x in (("keys + slice counts", us(a[:, 2] - a[:, 0])), ("first slice gathered, ranked, terms in LDS", us(a[:, 3] - a[:, 2])),
                ("its walk + the further slices", us(a[:, 7] - a[:, 3])), ("total", us(a[:, 7] - a[:, 0]))):
    print(f"{name:44s} p50 {np.median(x):7.1f} p90 {np.percentile(x, 90):7.1f} max {x.max():7.1f} us")
wg = a[:, 15]
print("guides per workgroup among the stamped:", np.bincount(np.bincount(wg)).tolist()[:6])
# where the slow first phases are: by start time inside the launch, by hits, by position in the workgroup's sequence
t0 = a[:, 0].min()
start = us(a[:, 0] - t0)
ph1 = us(a[:, 2] - a[:, 0])
print("launch span of the stamped guides: %.0f us" % us(a[:, 7].max() - t0))
for lo, hi in ((0, 50), (50, 200), (200, 400), (400, 800), (800, 1e9)):
    m = (start >= lo) & (start < hi)
    if m.any():
        print(f"  started at {lo:>4.0f}..{hi:<6.0f} us: {m.sum():5d} guides, first phase p50 {np.median(ph1[m]):6.1f} p90 {np.percentile(ph1[m], 90):6.1f}, total p50 {np.median(us(a[m, 7] - a[m, 0])):6.1f}")
for lo, hi in ((512, 768), (768, 1024), (1024, 1536), (1536, 2049)):
    m = (a[:, 1] >= lo) & (a[:, 1] < hi)
    if m.any():
        print(f"  hits {lo:>4d}..{hi:<4d}: {m.sum():5d} guides, first phase p50 {np.median(ph1[m]):6.1f} p90 {np.percentile(ph1[m], 90):6.1f}, total p50 {np.median(us(a[m, 7] - a[m, 0])):6.1f} p90 {np.percentile(us(a[m, 7] - a[m, 0]), 90):6.1f}")
