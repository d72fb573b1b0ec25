"""Analyse a rocprofv3 kernel-trace CSV of a multi-lane run: how many kernels are on the device at a time, per-queue busy
time, the largest gaps.   python tools/lanes_trace.py trace.csv [t_from_frac t_to_frac]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:60]) for r in rows]
t_lo, t_hi = min(e[0] for e in ev), max(e[1] for e in ev)
a, b = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
ev = [e for e in ev if e[0] >= a and e[1] <= b]
span = (b - a) / 1e3
print(f"window {span / 1e3:.2f} ms, {len(ev)} kernels")
pts = []
for s, e, q, n in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
depth, last, hist = 0, a, collections.Counter()
for t, d in pts:
    hist[depth] += t - last
    last = t
    depth += d
hist[depth] += b - last
tot = sum(hist.values())
print("kernels in flight: " + "  ".join(f"{k}: {100 * v / tot:.1f}%" for k, v in sorted(hist.items())))
byq = collections.defaultdict(float)
for s, e, q, n in ev:
    byq[q] += (e - s) / 1e3
print("busy per queue (ms): " + "  ".join(f"q{q}: {v / 1e3:.2f}" for q, v in sorted(byq.items())))
agg = collections.defaultdict(list)
for s, e, q, n in ev:
    agg[n].append((e - s) / 1e3)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  {n:60s} calls {len(v):5d} avg {sum(v) / len(v):8.1f} us  total {sum(v) / 1e3:8.2f} ms")
