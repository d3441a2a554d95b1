"""dev: mean counter values per dispatch of kernels matching a name, from rocprofv3 --pmc output directories
(usage: pmc_kernel.py NAME DIR [DIR ...])"""
import glob, sys
import pandas as pd
name = sys.argv[1]
for d in sys.argv[2:]:
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains(name)]
    print(d, "dispatches", df.Dispatch_Id.nunique())
    for k, v in df.groupby("Counter_Name").Counter_Value.mean().items():
        print(f"   {k:32s} {v:16.1f}")
