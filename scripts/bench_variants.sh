#!/bin/bash
# dev: time the 12 kbp bench with every variant library under build/var (compiler-flag experiments)
for lib in mythos_amd/lib/libmythos_hip.so build/var/lib_*.so mythos_amd/lib/libmythos_hip.so; do
  r=$(MYTHOS_HIP_LIB=$lib python bench.py --steps 3000 --warmup 300 --cpu-steps 0 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms']*1e3,2))")
  echo "${lib:-default} $r"
done
