#!/bin/bash
# round 4: the positions-only trajectory against no output and against the energy-trace route, 12 kbp, both precisions
set -e
mkdir -p gpurun_out/r04
for dt in f32 f64; do
  for se in 0 1; do
    python bench.py --dtype $dt --no-second-dtype --cpu-steps 0 --save-every $se > gpurun_out/r04/save_${dt}_${se}.json 2> gpurun_out/r04/save_${dt}_${se}.err
  done
  python bench.py --dtype $dt --no-second-dtype --cpu-steps 0 --save-every 1 --trace-energy > gpurun_out/r04/save_${dt}_1e.json 2>> gpurun_out/r04/save_${dt}_1e.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/save_*.json')):
    d=json.load(open(f)); print(f, round(d['value']), d['config']['timed_region'])
PY
