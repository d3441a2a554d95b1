#!/bin/bash
# dev (round 3): A/B of the variant libraries under build/var on the GPU box - 12 kbp fp32 + fp64, 100 kbp, 256 replicas.
# usage: scripts/exp_variants_r03.sh out.log [lib ...]
out=$1; shift
libs="$@"
[ -z "$libs" ] && libs="mythos_amd/lib/libmythos_hip.so build/var/lib_*.so mythos_amd/lib/libmythos_hip.so"
: > $out
for lib in $libs; do
  echo "== $lib" >> $out
  MYTHOS_HIP_LIB=$lib python bench.py --steps 2000 --warmup 200 --cpu-steps 0 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=d['config'].get('f64',{})
print('12kbp f32', round(d['value']), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), '| f64', round(f.get('steps_per_s',0)), 'kernel_us', round(f.get('kernel_ms',0)*1e3,2))" >> $out 2>&1
  MYTHOS_HIP_LIB=$lib python bench.py --bp 100000 --steps 300 --warmup 50 --cpu-steps 0 --no-second-dtype 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('100kbp f32', round(d['value']), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2))" >> $out 2>&1
  MYTHOS_HIP_LIB=$lib python scripts/bench_replicas.py 3000 2>/dev/null | tail -n 2 >> $out
done
cat $out
