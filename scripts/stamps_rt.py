"""(needs the diagnostic build: make -C mythos_amd/csrc clean && make -C mythos_amd/csrc DIAG=1)
Wall-clock view of the MD kernel from s_memrealtime stamps (MYTHOS_MD_ABLATE=384): 10 ns ticks.
The dump holds the last two launches (even step | odd step)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(2, -1, 4, 8).astype(np.int64)
if a[0, :, :, 0].min() > a[1, :, :, 0].min():
    a = a[::-1]
for k in range(2):
    st, en = a[k, :, :, 0], a[k, :, :, 7]
    print("launch %d: first start %.2f  last start %.2f  first end %.2f  last end %.2f (us rel. first start of launch 0)" % (
        k, *(np.array([st.min(), st.max(), en.min(), en.max()]) - a[0, :, :, 0].min()) / 100))
d = (a[:, :, :, 7] - a[:, :, :, 0]) / 100
print("block duration us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(d, [10, 50, 90, 100])))
print("gap last end(0) -> first start(1): %.2f us;  period %.2f us" % (
    (a[1, :, :, 0].min() - a[0, :, :, 7].max()) / 100, (a[1, :, :, 0].min() - a[0, :, :, 0].min()) / 100))
seg = np.diff(a, axis=3) / 100
print("segments us (median):", np.round(np.median(seg, axis=(0, 1, 2)), 2))
