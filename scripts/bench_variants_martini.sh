#!/bin/bash
# dev: MARTINI bench with every variant library under build/var
for lib in mythos_amd/lib/libmythos_hip.so build/var/lib_*.so mythos_amd/lib/libmythos_hip.so; do
  r=$(MYTHOS_HIP_LIB=$lib python bench.py --workload martini-bilayer --cpu-steps 0 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']))")
  echo "$lib $r"
done
