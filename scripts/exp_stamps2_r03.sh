#!/bin/bash
# dev (round 3): wall-clock phases of the step kernel for two diagnostic builds, at a given size
bp=${1:-100000}
for lib in build/var/lib_diag0.so build/var/lib_diag1.so; do
  echo "== $lib bp $bp"
  MYTHOS_HIP_LIB=$lib MYTHOS_MD_ABLATE=384 MYTHOS_MD_STAMPS=gpurun_out/st.bin python bench.py --bp $bp --steps 100 --warmup 30 --cpu-steps 0 --no-second-dtype --repeats 1 > /dev/null 2>&1
  python scripts/stamps_rt.py gpurun_out/st.bin
  MYTHOS_HIP_LIB=$lib MYTHOS_MD_ABLATE=128 MYTHOS_MD_STAMPS=gpurun_out/st_cyc.bin python bench.py --bp $bp --steps 100 --warmup 30 --cpu-steps 0 --no-second-dtype --repeats 1 > /dev/null 2>&1
  python scripts/stamps_roles.py gpurun_out/st_cyc.bin
done
