#!/bin/bash
# round 4, VERDICT item 4: bounds first.  (c) kernel-argument size against the launch boundary (build/exp_kernarg from
# scripts/exp_kernarg_r04.hip); (a) the radial pass with every second iteration left out and (b) acos + rsqrt of the angular
# modulations replaced by two multiply-adds - both WRONG PHYSICS, variant libraries from scripts/build_variant.sh,
# alternating with the product on one box, 12 kbp fp32, 2 000 steps after 200.
mkdir -p gpurun_out/r04
build/exp_kernarg | tee gpurun_out/r04/kernarg.txt
for round in 1 2; do
  for v in "" build/var/lib_halfrad.so build/var/lib_meshb.so; do
    if [ -n "$v" ]; then export MYTHOS_HIP_LIB=$v; else unset MYTHOS_HIP_LIB; fi
    python bench.py --no-second-dtype --no-secondary --cpu-steps 0 --steps 2000 --warmup 200 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('${v:-product}', round(d['value']), 'steps/s  kernel', round(1e3*d['roofline']['kernel_ms'],2), 'us', d['config']['neighbor_list']['out_of_turn_rebuilds'])"
  done
done 2>&1 | tee gpurun_out/r04/md_bounds.txt
