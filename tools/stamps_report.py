"""Summarise a wave-stamp dump of the diagnostic build (librt built with -DRT_STAMPS):
    RT_STAMPS_FILE=stamps.bin RT_PIPES=1 python bench.py ... ; python tools/stamps_report.py stamps.bin"""
import sys

import numpy as np

SLOTS, WAVES = 32, 16384
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(SLOTS, WAVES, 4)
for s in range(SLOTS):
    t0, t1, hw, x = (a[s, :, k] for k in range(4))
    live = t1 > 0
    if not live.any():
        continue
    t0, t1, hw, x = t0[live], t1[live], hw[live], x[live]
    rays = (x & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((x >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    hw = hw.astype(np.int64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    unit = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    begin = int(t0.min())
    dur = (t1 - t0).astype(np.int64) / 100.0          # us (100 MHz counter)
    start = (t0.astype(np.int64) - begin) / 100.0
    end = (t1.astype(np.int64) - begin) / 100.0
    span = end.max()
    # average number of resident waves per SIMD over the launch
    n_simd = len(np.unique(unit * 4 + simd))
    occ = dur.sum() / span / max(n_simd, 1)
    print(f"launch {s}: {live.sum()} waves on {len(np.unique(unit))} CUs / {n_simd} SIMDs, span {span:.0f} us, "
          f"wave life mean {dur.mean():.0f} us (min {dur.min():.0f}, max {dur.max():.0f}), rays/wave {rays.mean():.0f}, "
          f"starts: 50% by {np.percentile(start, 50):.0f} us, 90% by {np.percentile(start, 90):.0f} us, last {start.max():.0f} us; "
          f"ends: 10% by {np.percentile(end, 10):.0f}, 50% by {np.percentile(end, 50):.0f}; mean resident waves/SIMD {occ:.2f}")
    per_simd = np.bincount(unit * 4 + simd)
    per_simd = per_simd[per_simd > 0]
    print(f"          waves per SIMD: min {per_simd.min()} max {per_simd.max()}; waves per XCC: {np.bincount(xcc).tolist()}")

# resident waves per SIMD over time, all stamped launches together (they overlap when the renderer
# runs more than one pipeline)
allw = a.reshape(-1, 4)
allw = allw[allw[:, 1] > 0]
t0 = allw[:, 0].astype(np.int64)
t1 = allw[:, 1].astype(np.int64)
lo, hi = t0.min(), t1.max()
edges = np.linspace(lo, hi, 41)
n_simd_all = 1024
print("resident stamped waves per SIMD over time (40 slices of %.0f us):" % ((hi - lo) / 100.0 / 40))
print(" ".join("%.1f" % (((t0 <= m) & (t1 > m)).sum() / n_simd_all) for m in 0.5 * (edges[:-1] + edges[1:])))
