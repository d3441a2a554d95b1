#!/bin/bash
# dev: energy bench with every variant library under build/var
for lib in mythos_amd/lib/libmythos_hip.so build/var/lib_*.so mythos_amd/lib/libmythos_hip.so; do
  echo "== $lib"
  MYTHOS_HIP_LIB=$lib python scripts/bench_energy.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['bp'], d['dtype'], [round(d[k]['ms_per_call'], 3) for k in ('energy', 'energy+forces', 'energy+forces+dU/dtheta')])
"
done
