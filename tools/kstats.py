"""Summarise a rocprofv3 kernel-trace CSV: per kernel (truncated name) calls, avg us, total ms."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    n = n.split("(")[0][:90]
    agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{n:90s} calls {len(v):5d} avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:8.2f} ms ({100*sum(v)/tot:.1f}%)")
