#!/bin/bash
# round 4, VERDICT item 4(a): what a radial pass of ONE iteration per segment could leave (16 lanes per nucleotide).
# WRONG PHYSICS, short samples from the ideal helix: "halfrad" leaves out every second iteration of both radial loops - and
# with them the angular items those iterations would have flagged; "halfflag" keeps the radial pass whole and drops only
# those items.  The radial chain's share = kernel(halfflag) - kernel(halfrad).  Variant libraries: scripts/build_variant.sh
# {halfrad,halfflag} langevin.hip -DMYTHOS_MD_EXP_HALF_{RADIAL,FLAGS}.
for round in 1 2; do
  for v in "" build/var/lib_halfrad.so build/var/lib_halfflag.so; do
    if [ -n "$v" ]; then export MYTHOS_HIP_LIB=$v; else unset MYTHOS_HIP_LIB; fi
    python bench.py --no-second-dtype --no-secondary --cpu-steps 0 --steps 100 --warmup 0 --min-warmup-ms 0 --repeats 3 --instrument-steps 64 2>gpurun_out/r04/halfrad.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('${v:-product}', round(d['value']), 'steps/s  kernel', round(1e3*d['roofline']['kernel_ms'],2), 'us', d['config']['neighbor_list']['mean_row'])" || tail -3 gpurun_out/r04/halfrad.err
  done
done 2>&1 | tee gpurun_out/r04/md_bounds2.txt
