#!/bin/bash
# dev (round 3): wall-clock phases of the step kernel (diagnostic build build/var/lib_diag.so: -DMYTHOS_MD_DIAG)
MYTHOS_HIP_LIB=build/var/lib_diag.so MYTHOS_MD_ABLATE=384 MYTHOS_MD_STAMPS=gpurun_out/r03_stamps.bin python bench.py --steps 300 --warmup 100 --cpu-steps 0 --no-second-dtype --repeats 1 > /dev/null 2>&1
python scripts/stamps_rt.py gpurun_out/r03_stamps.bin
python scripts/stamps_tail.py gpurun_out/r03_stamps.bin 2>/dev/null | tail -12
MYTHOS_HIP_LIB=build/var/lib_diag.so MYTHOS_MD_ABLATE=128 MYTHOS_MD_STAMPS=gpurun_out/r03_stamps_cyc.bin python bench.py --steps 300 --warmup 100 --cpu-steps 0 --no-second-dtype --repeats 1 > /dev/null 2>&1
python scripts/stamps_roles.py gpurun_out/r03_stamps_cyc.bin
