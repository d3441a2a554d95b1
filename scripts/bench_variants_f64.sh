# dev: fp64 step rate at the driver's 20-step command and at 1 500 steps, for every variant library under build/var
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 100 --warmup 5" "--steps 1500 --warmup 200"; do
for lib in mythos_amd/lib/libmythos_hip.so build/var/lib_*.so; do
  r=$(MYTHOS_HIP_LIB=$lib python bench.py --dtype f64 $args --cpu-steps 0 --no-second-dtype 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1e3,2), round(d['roofline']['kernel_ms']*1e3,2), d['config']['neighbor_list'].get('out_of_turn_rebuilds'))")
  echo "$args $lib $r"
done; done
