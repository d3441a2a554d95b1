#!/bin/bash
# dev (round 3): the driver-style 20-step call (five samples) and a 2 000-step call, per library
out=$1; shift
: > $out
for lib in "$@"; do
  echo "== $lib" >> $out
  for rep in 1 2; do
  MYTHOS_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-second-dtype --repeats 9 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('20 steps:', round(d['value']), [round(x,3) for x in d['config']['samples_ms']], d['config']['scheduled_rebuilds_per_sample'])" >> $out 2>&1
  done
  MYTHOS_HIP_LIB=$lib python bench.py --steps 2000 --warmup 200 --cpu-steps 0 --no-second-dtype --repeats 3 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('2000 steps:', round(d['value']), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2))" >> $out 2>&1
done
cat $out
