#!/bin/bash
# dev (round 3): the slope of the step time in bytes written - a variant that writes 32 B MORE per
# nucleotide and step (a second copy of a1 / a3 into a side buffer; right physics) against the product, alternating.
# usage: scripts/exp_writemore_r03.sh out.log product.so variant.so
out=$1; shift
: > $out
for lib in $1 $2 $1 $2; do
  echo "== $lib" >> $out
  for bp in 12000 100000; do
    steps=2000; [ $bp = 100000 ] && steps=300
    MYTHOS_HIP_LIB=$lib python bench.py --bp $bp --steps $steps --warmup 100 --cpu-steps 0 --no-second-dtype --repeats 3 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$bp f32 steps/s', round(d['value']), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'recoveries', d['config'].get('neighbor_list',{}).get('out_of_turn_rebuilds'))" >> $out 2>&1
  done
done
cat $out
