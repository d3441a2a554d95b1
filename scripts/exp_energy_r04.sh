#!/bin/bash
# round 4: fp64 energy kernel variants on the DiffTRe shape (6 400 frames x 64 nt): product, park from MODE 1 (B), + MODE 1 at three
# workgroups per CU (C), park from MODE 0 (D), + MODE 0 at four per CU (E)
for v in "" build/var/lib_eB.so build/var/lib_eC.so build/var/lib_eD.so build/var/lib_eE.so; do
  echo "== lib: ${v:-product}"
  if [ -n "$v" ]; then export MYTHOS_HIP_LIB=$v; else unset MYTHOS_HIP_LIB; fi
  python scripts/bench_energy.py --difftre 2>&1 | grep float64 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print([round(d[k]['ms_per_call'],3) for k in ('energy','energy+forces','energy+forces+dU/dtheta')])"
done
