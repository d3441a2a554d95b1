# needs the diagnostic build: make -C mythos_amd/csrc clean && make -C mythos_amd/csrc DIAG=1
"""Angular-pass duration per wave role from a cycle-stamp dump (role = (wave + workgroup) & 3)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(2, -1, 4, 8).astype(np.int64)
a = a[0] if a[0, :, :, 0].min() > 0 else a[1]
nb = a.shape[0]
d = np.diff(a, axis=2)
role = (np.arange(4)[None, :] + np.arange(nb)[:, None]) & 3
for r, name in enumerate(["bonded", "base-pair A", "base-pair B", "coaxial"]):
    v, w = d[:, :, 2][role == r], d[:, :, 3][role == r]
    print(f"{name:12s} angular pass p10 {np.percentile(v,10):7.0f} p50 {np.percentile(v,50):7.0f} p90 {np.percentile(v,90):7.0f} | wait at the barrier p50 {np.percentile(w,50):6.0f}")
