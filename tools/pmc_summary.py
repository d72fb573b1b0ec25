"""Average rocprofv3 --pmc counter values per dispatch for kernels whose name contains a pattern."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
agg = collections.defaultdict(list)
for r in rows:
    if pat in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"  {k:32s} n={len(v):4d} avg {sum(v)/len(v):14.1f}")
