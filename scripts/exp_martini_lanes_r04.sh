#!/bin/bash
# round 4: lanes per bead (MYTHOS_MM_G) and row entries per lane in flight (MYTHOS_MM_BATCH) of martini_md_step_kernel,
# 20 480 beads; variant libraries from scripts/build_variant.sh mmG<g>B<b> martini_md.hip "-DMYTHOS_MM_G=<g> -DMYTHOS_MM_BATCH=<b>"
mkdir -p gpurun_out/r04
for round in 1 2; do
  for v in "" build/var/lib_mmG16B4.so build/var/lib_mmG16B3.so build/var/lib_mmG16B2.so build/var/lib_mmG32B2.so; do
    if [ -n "$v" ]; then export MYTHOS_HIP_LIB=$v; else unset MYTHOS_HIP_LIB; fi
    python bench.py --workload martini-bilayer --cpu-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('${v:-product (8 lanes, batch 4)}', round(d['value']), 'steps/s  kernel', round(1e3*d['roofline']['kernel_ms'],2), 'us  loop', round(1e3*d['roofline']['loop_ms_per_launch'],2))"
  done
done 2>&1 | tee gpurun_out/r04/martini_lanes.txt
