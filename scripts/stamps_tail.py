"""(diagnostic build) Per-wave wall-clock stamps of the slowest workgroups of the first dumped launch (us)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(2, -1, 4, 8).astype(np.int64)
if a[0,:,:,0].min() > a[1,:,:,0].min(): a = a[::-1]
L = a[int(sys.argv[2]) if len(sys.argv) > 2 else 0]
t0 = L[:,:,0].min()
st = (L[:,:,0].min(1)-t0)/100; en = (L[:,:,7].max(1)-t0)/100
print("end percentiles 50/90/99/100:", np.round(np.percentile(en,[50,90,99,100]),2))
for b in list(np.argsort(-en)[:4]) + list(np.argsort(en)[:2]):
    print("block", b, "start", round(st[b],2), "end", round(en[b],2))
    for w in range(4):
        print("   wave", w, "role", (w + b) & 3, np.round((L[b,w]-t0)/100 - st[b],2).tolist())
