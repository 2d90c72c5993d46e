"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch of kernels matching a pattern."""
import csv, glob, sys, collections
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "attn_fwd")
acc = collections.defaultdict(list)
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
