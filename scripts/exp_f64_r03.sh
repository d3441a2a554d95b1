#!/bin/bash
# dev (round 3): fp64 A/B of variant libraries - 12 kbp and 100 kbp.   usage: scripts/exp_f64_r03.sh out.log lib...
out=$1; shift
: > $out
for lib in "$@"; do
  echo "== $lib" >> $out
  for bp in 12000 100000; do
    steps=1500; [ $bp = 100000 ] && steps=200
    MYTHOS_HIP_LIB=$lib python bench.py --bp $bp --dtype f64 --steps $steps --warmup 100 --cpu-steps 0 --no-second-dtype --repeats 3 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$bp f64', round(d['value']), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2))" >> $out 2>&1
  done
done
cat $out
