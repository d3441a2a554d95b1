#!/bin/bash
# round 4: the MARTINI loop with the classic 27-cell row builder (mm_subcells=1) against cells of half the list range
# (the product), alternating on one box, then the kernel trace of each
mkdir -p gpurun_out/r04
for round in 1 2; do
  for v in 1 0; do
    python bench.py --workload martini-bilayer --cpu-steps 0 --debug-set mm_subcells=$v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('mm_subcells=$v', round(d['value']), 'steps/s  kernel', round(1e3*d['roofline']['kernel_ms'],2), 'us  loop', round(1e3*d['roofline']['loop_ms_per_launch'],2), d['config']['neighbor_list']['mean_row'])"
  done
done 2>&1 | tee gpurun_out/r04/martini_ab.txt
export TMPDIR=/tmp
for v in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/mm_trace_$v -- python bench.py --workload martini-bilayer --cpu-steps 0 --steps 500 --warmup 100 --repeats 1 --debug-set mm_subcells=$v > gpurun_out/r04/mm_trace_$v.log 2>&1
  f=$(find gpurun_out/r04/mm_trace_$v -name '*kernel_stats.csv' | head -n 1)
  echo "== mm_subcells=$v"; head -8 "$f" | cut -c1-150
  cp "$f" gpurun_out/r04/mm_kernel_stats_$v.csv; rm -rf gpurun_out/r04/mm_trace_$v
done 2>&1 | tee -a gpurun_out/r04/martini_ab.txt
