#!/bin/bash
# needs the diagnostic build: make -C mythos_amd/csrc clean && make -C mythos_amd/csrc DIAG=1
# dev: dynamic VALU instructions per wavefront of md_step_kernel under the kernel's ablation bits
# (MYTHOS_MD_ABLATE: 1 no neighbour rows, 2 no angular pass, 8 no unbonded angular lists, 16 no bonded items, 4 no integration)
export TMPDIR=/tmp
for a in 0 1 2 8 16 4 3 7; do
  rm -rf gpurun_out/vb_$a
  MYTHOS_MD_ABLATE=$a rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/vb_$a -- python bench.py --steps 20 --warmup 0 --cpu-steps 0 --rebuild-every 10 --skin 1.0 > gpurun_out/vb_$a.log 2>&1
  python - $a <<'PY'
import sys, glob
import pandas as pd
a = sys.argv[1]
f = glob.glob(f"gpurun_out/vb_{a}/**/*counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df = df[df.Kernel_Name.str.contains("md_step_kernel")]
m = df.groupby("Counter_Name").Counter_Value.median()
print(f"ablate={a:>2}: VALU/wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:8.1f}   lanes/inst {m['SQ_THREAD_CYCLES_VALU'] / (64 * m['SQ_ACTIVE_INST_VALU']):.2f}")
PY
done
