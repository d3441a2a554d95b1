#!/bin/bash
# dev (round 3): observables in the epilogue of the energy launch (OBS instantiations) against the stand-alone
# observables launch queued behind the plain energy launch; wall clock per call, DiffTRe shape, alternating.
# usage: scripts/exp_obsfuse_r03.sh out.log product.so variant.so
out=$1; shift
: > $out
for lib in $1 $2 $1 $2; do
  echo "== $lib" >> $out
  MYTHOS_HIP_LIB=$lib python scripts/bench_energy.py --obs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['dtype'], {k:round(v['ms_per_call'],3) for k,v in d.items() if isinstance(v,dict)})" >> $out 2>&1
done
MYTHOS_HIP_LIB=$1 python scripts/bench_energy.py --difftre 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('plain', d['dtype'], {k:round(v['ms_per_call'],3) for k,v in d.items() if isinstance(v,dict)})" >> $out 2>&1
cat $out
