#!/usr/bin/env python3
"""Digest of an ISSL_SCAN_STAMPS dump (4 u64 per scan wave: start, end [100 MHz ticks], XCC_ID<<32|HW_ID, tiles taken
in the low half and ticks spent waiting for tile planes in the high half).

    ISSL_SCAN_STAMPS=/tmp/st.bin python tools/quick_perf.py ... ; python tools/scan_stamps.py /tmp/st.bin
"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
# the scan waves come first (16 per workgroup, 1024 workgroups unless scan_blocks says otherwise: second argument); the
# dump also holds the phase clocks of the many-hit replay behind them
a = a[:int(sys.argv[2]) if len(sys.argv) > 2 else 16384]
wave = np.arange(len(a))
live = a[:, 1] > 0
a, wave = a[live], wave[live]
t0 = a[:, 0].min()
s = (a[:, 0] - t0) / 100.0
e = (a[:, 1] - t0) / 100.0
hw = a[:, 2] & np.uint64(0xFFFFFFFF)
xcc = (a[:, 2] >> np.uint64(32)) & np.uint64(0xF)
cu = (hw >> np.uint64(8)) & np.uint64(0xF)
sh = (hw >> np.uint64(12)) & np.uint64(1)
se = (hw >> np.uint64(13)) & np.uint64(7)
simd = (hw >> np.uint64(4)) & np.uint64(3)
units = (a[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
plane_wait = (a[:, 3] >> np.uint64(32)).astype(np.float64) / 100.0  # us
pc = lambda x, q: np.percentile(x, q).round(1)
print(f"waves {len(a)}  start max {s.max():.1f} us | end min {e.min():.1f} p10 {pc(e,10)} p50 {pc(e,50)} p90 {pc(e,90)} max {e.max():.1f} us")
print(f"tiles per wave: min {units.min()} p10 {pc(units,10)} p50 {pc(units,50)} p90 {pc(units,90)} max {units.max()}  total {units.sum()}")
wg = wave // 16
wg_end = np.array([e[wg == g].max() for g in np.unique(wg)])
wg_first = np.array([e[wg == g].min() for g in np.unique(wg)])
print(f"workgroups {len(wg_end)}: end min {wg_end.min():.1f} p10 {pc(wg_end,10)} p50 {pc(wg_end,50)} p90 {pc(wg_end,90)} max {wg_end.max():.1f} | "
      f"spread inside a workgroup (last - first wave end) p50 {pc(wg_end - wg_first,50)} max {(wg_end - wg_first).max():.1f}")
cuid = (xcc.astype(np.int64) * 8 + se.astype(np.int64)) * 32 + sh.astype(np.int64) * 16 + cu.astype(np.int64)
ids = np.unique(cuid)
cu_end = np.array([e[cuid == c].max() for c in ids])
cu_first = np.array([wg_end[np.unique(wg[cuid == c])].min() for c in ids])
cu_waves = np.array([(cuid == c).sum() for c in ids])
cu_units = np.array([units[cuid == c].sum() for c in ids])
print(f"CUs seen {len(ids)} (waves per CU min {cu_waves.min()} max {cu_waves.max()}): end min {cu_end.min():.1f} p10 {pc(cu_end,10)} p50 {pc(cu_end,50)} "
      f"p90 {pc(cu_end,90)} max {cu_end.max():.1f} | first workgroup of the CU done p50 {pc(cu_first,50)}")
print(f"tiles per CU: min {cu_units.min()} p50 {pc(cu_units,50)} max {cu_units.max()}")
for x in np.unique(xcc):
    m = xcc == x
    print(f"  xcc {int(x)}: waves {m.sum()} end p50 {pc(e[m],50)} max {e[m].max():.1f}  tiles {units[m].sum()}")
# occupancy over time: how many waves are still running at t
for q in (0.5, 0.7, 0.8, 0.9, 0.95):
    t = e.max() * q
    print(f"  waves still running at {q:.0%} of the kernel ({t:.0f} us): {(e > t).sum()}")
# wave-slot time lost between a wave's own end and the end of its workgroup (the slots of a 16-wave workgroup are only
# re-usable by the next workgroup when all of them are free), and between a workgroup's end and the kernel's end
wg_ids = np.unique(wg)
wg_end_of = dict(zip(wg_ids, wg_end))
lost_in_wg = sum((wg_end_of[g] - e[wg == g]).sum() for g in wg_ids)
busy = (e - s).sum()
span = e.max() * 8192.0  # wave slots of the chip x kernel duration
print(f"wave time: busy {busy / span:.1%} of (8192 slots x kernel time); waiting for the workgroup's last wave {lost_in_wg / span:.1%}; "
      f"rest (ramp, tail, gaps between workgroups) {1 - (busy + lost_in_wg) / span:.1%}")
per_tile = (e - s) / np.maximum(units, 1)
print(f"time per tile and wave: p10 {pc(per_tile,10)} p50 {pc(per_tile,50)} p90 {pc(per_tile,90)} us")
wait_per_tile = plane_wait / np.maximum(units, 1)
print(f"waiting for tile planes, per tile and wave: p10 {pc(wait_per_tile,10)} p50 {pc(wait_per_tile,50)} p90 {pc(wait_per_tile,90)} us; "
      f"{plane_wait.sum() / busy:.1%} of the waves' busy time")
