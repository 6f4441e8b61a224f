#!/usr/bin/env python3
"""Digest of the big-guide phase clocks in an ISSL_SCAN_STAMPS dump (k_replay_big writes 16 u64 per guide after the
scan's records: start, hits, after partition, slice-0 start, its length, after its sort, after its terms, end, kept).

    ISSL_SCAN_STAMPS=/tmp/st.bin python bench.py --dist markov --steps 5 --warmup 3 --no-cpu-baseline
    python tools/replay_stamps.py /tmp/st.bin [1024]      (the 256-thread build's guides, or the 1024-thread build's)
"""
import numpy as np, sys
base = 524288 + (131072 if len(sys.argv) > 2 and sys.argv[2] == '1024' else 65536)  # issl_device.hpp: kStampsBig1024 / kStampsBig256
a=np.fromfile(sys.argv[1],dtype=np.uint64)[base:base+16*4096].reshape(-1,16).astype(np.int64)
a=a[a[:,0]>0]
t0=a[:,0].min()
print("guides",len(a))
us=lambda x: x/100.0
part=us(a[:,2]-a[:,0]); pre=us(a[:,3]-a[:,2]); sort=us(a[:,5]-a[:,3]); terms=us(a[:,6]-a[:,5]); rest=us(a[:,7]-a[:,6]); tot=us(a[:,7]-a[:,0])
for name,x in (("partition",part),("sort slice0",sort),("terms slice0",terms),("walk+other slices",rest),("total",tot)):
    print(f"{name:18s} p50 {np.median(x):7.1f} p90 {np.percentile(x,90):7.1f} max {x.max():7.1f} us")
print("h p50",np.median(a[:,1]),"len0 p50",np.median(a[:,4]),"kept p50",np.median(a[:,8]))
print("start spread us: p50",np.median(us(a[:,0]-t0)),"max",us(a[:,0]-t0).max(), " end max", us(a[:,7]-t0).max())
blk=a[:,15]
print("guides per block max", np.bincount(blk).max())
