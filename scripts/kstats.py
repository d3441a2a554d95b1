"""dev: top kernels of a rocprofv3 --kernel-trace --stats run directory (usage: kstats.py DIR [N])"""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), f'{float(r["AverageNs"]) / 1e3:9.2f} us', r["Percentage"].rjust(7))
