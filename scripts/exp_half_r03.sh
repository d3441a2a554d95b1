#!/bin/bash
# dev (round 3): the pair-once bound (MYTHOS_MD_EXP_HALF_ITEMS: wrong physics, one base-pair sweep idle) where the chip is
# issue-bound - 100 kbp and 256 replicas - in runs short enough that the broken force balance has not moved anything yet
out=$1
: > $out
for lib in mythos_amd/lib/libmythos_hip.so build/var_half/lib_half.so build/var/lib_lean1.so build/var_half/lib_lean1half.so mythos_amd/lib/libmythos_hip.so; do
  echo "== $lib" >> $out
  for bp in 12000 100000; do
    MYTHOS_HIP_LIB=$lib python bench.py --bp $bp --steps 40 --warmup 5 --cpu-steps 0 --no-second-dtype 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$bp f32 steps/s', round(d['value']), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'recoveries', d['config'].get('neighbor_list',{}).get('out_of_turn_rebuilds'))" >> $out 2>&1
  done
  MYTHOS_HIP_LIB=$lib python scripts/bench_replicas.py 100 2>/dev/null | tail -n 2 >> $out
done
cat $out
