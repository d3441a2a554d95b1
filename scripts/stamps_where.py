"""(diagnostic build, MYTHOS_MD_ABLATE=896 = stamps | realtime | where) What makes the slow workgroups of a launch slow:
duration of every workgroup against the number of workgroups that shared its CU, its XCD, its angular work and its
start time.  Slot 5 = length of the wavefront's angular work list, slot 6 = XCC_ID << 32 | HW_ID."""
import sys
from collections import Counter

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(2, -1, 4, 8)
t = a[..., [0, 1, 2, 3, 4, 7]].astype(np.int64)
if t[0, :, :, 0].min() > t[1, :, :, 0].min():
    a, t = a[::-1], t[::-1]
for k in range(2):
    L, T = a[k], t[k]
    t0 = T[:, :, 0].min()
    st, en = (T[:, :, 0].min(1) - t0) / 100, (T[:, :, 5].max(1) - t0) / 100
    dur = en - st
    hw = L[:, 0, 6]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    h = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cu, sh, se = (h >> 8) & 0xF, (h >> 12) & 1, (h >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = Counter(cuid.tolist())
    share = np.array([per_cu[c] for c in cuid])
    items = L[:, :, 5].astype(np.int64)  # per wavefront (role order rotates with the workgroup)
    print(f"launch {k}: {len(dur)} workgroups on {len(per_cu)} CUs; workgroups per CU:", dict(sorted(Counter(per_cu.values()).items())))
    print("  duration us p10/p50/p90/max: %.2f %.2f %.2f %.2f; last end %.2f" % (*np.percentile(dur, [10, 50, 90, 100]), en.max()))
    for s_ in sorted(set(share.tolist())):
        m = share == s_
        print(f"  {s_} per CU: n={m.sum():4d}  start {st[m].mean():5.2f}  duration mean {dur[m].mean():5.2f} max {dur[m].max():5.2f}  end max {en[m].max():5.2f}")
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"  XCD {x}: n={m.sum():3d} CUs {len(set(cuid[m].tolist())):2d} duration mean {dur[m].mean():5.2f} end max {en[m].max():5.2f}")
    tot = items.sum(1)
    print("  angular items per workgroup p10/p50/p90/max:", np.percentile(tot, [10, 50, 90, 100]).tolist(),
          " longest list of a wavefront p50/max:", np.percentile(items.max(1), [50, 100]).tolist())
    print("  corr(duration, items) %.2f  corr(duration, longest list) %.2f  corr(duration, share) %.2f  corr(duration, start) %.2f" % (
        np.corrcoef(dur, tot)[0, 1], np.corrcoef(dur, items.max(1))[0, 1], np.corrcoef(dur, share)[0, 1], np.corrcoef(dur, st)[0, 1]))
    slow = np.argsort(-en)[:6]
    for b in slow:
        seg = np.diff((T[b] - t0) / 100, axis=1)
        print(f"   slow wg {b}: xcd {xcc[b]} cu {cuid[b]} share {share[b]} start {st[b]:.2f} end {en[b]:.2f} items {items[b].tolist()} "
              f"phases(radial,gap,angular,gap,rest) per wave {np.round(seg, 2).tolist()}")
