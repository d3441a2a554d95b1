# dev: fp64 / fp32 step rate of every variant library under build/var at 12 kbp and 100 kbp
for args in "--dtype f64 --bp 12000" "--dtype f32 --bp 12000" "--dtype f64 --bp 100000" "--dtype f32 --bp 100000"; do
for lib in mythos_amd/lib/libmythos_hip.so build/var/lib_*.so mythos_amd/lib/libmythos_hip.so; do
  r=$(MYTHOS_HIP_LIB=$lib python bench.py $args --steps 1500 --warmup 200 --cpu-steps 0 --no-second-dtype 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms']*1e3,2))")
  echo "$args $lib $r"
done; done
