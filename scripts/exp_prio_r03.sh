#!/bin/bash
# dev (round 3): wave-priority maps of the step kernel (build/var/lib_p*.so) - 12 kbp fp32, 100 kbp, 256 replicas.
# usage: scripts/exp_prio_r03.sh out.log lib...
out=$1; shift
: > $out
for lib in "$@"; do
  a=$(MYTHOS_HIP_LIB=$lib python bench.py --steps 2000 --warmup 200 --cpu-steps 0 --no-second-dtype --repeats 3 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']))")
  b=$(MYTHOS_HIP_LIB=$lib python bench.py --bp 100000 --steps 300 --warmup 50 --cpu-steps 0 --no-second-dtype --repeats 3 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']))")
  c=$(MYTHOS_HIP_LIB=$lib python scripts/bench_replicas.py 3000 2>/dev/null | tail -n 1 | awk "{print \$7}")
  echo "$lib 12kbp $a 100kbp $b 256rep $c" >> $out
done
cat $out
