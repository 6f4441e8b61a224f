#!/usr/bin/env python3
"""Digest of k_replay_mid's phase clocks in an ISSL_SCAN_STAMPS dump (16 u64 per listed guide behind the scan's records;
entries with [14] == 1 are k_replay_mid's).  python tools/replay_mid_stamps.py /tmp/st.bin"""
import numpy as np, sys
a = np.fromfile(sys.argv[1], dtype=np.uint64)[65536:65536 + 16 * 4096].reshape(-1, 16).astype(np.int64)
a = a[(a[:, 0] > 0) & (a[:, 14] == 1)]
us = lambda x: x / 100.0
print("mid guides stamped", len(a), "h p50/p90", np.percentile(a[:, 1], [50, 90]), "head cnt p50", np.median(a[:, 4]), "kept p50/p90", np.percentile(a[:, 8], [50, 90]),
      "took the head path", (a[:, 9] > 0).mean(), "left inside the head", ((a[:, 9] > 0) & (a[:, 8] < a[:, 9])).mean())
head = a[a[:, 9] > 0]
for name, x in (("load keys + first slice", us(a[:, 2] - a[:, 0])), ("range + groups", us(a[:, 3] - a[:, 2])),
                ("gather + rank (head)", us(head[:, 5] - head[:, 3])), ("walk head", us(head[:, 6] - head[:, 5])),
                ("rest (full sort + walk)", us(head[:, 7] - head[:, 6])), ("total", us(a[:, 7] - a[:, 0]))):
    print(f"{name:26s} p50 {np.median(x):7.1f} p90 {np.percentile(x, 90):7.1f} max {x.max():7.1f} us")
