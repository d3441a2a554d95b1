# needs the diagnostic build: make -C mythos_amd/csrc clean && make -C mythos_amd/csrc DIAG=1
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 8).astype(np.int64)
d = np.diff(a[:, :, :8], axis=2)  # (blocks, waves, 7)
names = ["phase1", "barrier1-2", "phase2", "barrier3", "fold", "barrier4", "integrate+store"]
print("cycles per segment, median over blocks, per wave index (0 = bonded wave)")
for k, n in enumerate(names):
    print(f"{n:18s}", [int(np.median(d[:, w, k])) for w in range(4)])
tot = a[:, :, 7] - a[:, :, 0]
print("total             ", [int(np.median(tot[:, w])) for w in range(4)])
print("kernel span (first start -> last end):", int(a[:, :, 7].max() - a[:, :, 0].min()))
