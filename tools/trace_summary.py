"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals for the LAST enhance call in the trace, and,
given the OU_TRACE log of the same run, the per-layer duration / achieved TFLOP/s of one score forward."""
import csv
import re
import sys
from collections import defaultdict


def main(trace_csv, ou_trace_log=None):
    rows = list(csv.DictReader(open(trace_csv)))
    starts = [i for i, r in enumerate(rows) if "pad_normalize" in r["Kernel_Name"]]
    seg = rows[starts[-1]:] if starts else rows
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    tot = defaultdict(lambda: [0, 0])
    for r in seg:
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("ou::", "")
        if len(n) > 60:
            n = n[:57] + "..."
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tot[n][0] += d
        tot[n][1] += 1
    busy = sum(v[0] for v in tot.values())
    print(f"last enhance: span {(t1-t0)/1e6:.3f} ms, kernel-busy {busy/1e6:.3f} ms, {len(seg)} dispatches")
    for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print(f"  {n:60s} {c:5d} calls {d/1e3:10.1f} us  {100*d/busy:5.1f}%  avg {d/c/1e3:8.1f} us")
    if ou_trace_log:
        names = [l.split() for l in open(ou_trace_log) if l.startswith("OU_TRACE conv") or l.startswith("OU_TRACE chain")]
        # conv launches of the last enhance, in order, align with the last `per` trace lines
        last = [r for r in seg if any(k in r["Kernel_Name"] for k in ("conv_mfma_kernel", "conv_chain_kernel",
                                                                      "conv_direct", "rate_down_kernel", "rate_up_kernel"))]
        lines = names[-len(last):]
        print(f"per-layer (last enhance, {len(last)} conv launches):")
        seen = set()
        for r, l in zip(last, lines):
            nm = l[2]
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if l[1] == "chain":
                key = nm
                if key in seen and not nm.startswith("cond."):
                    continue
                seen.add(key)
                print(f"  {nm:28s} {' '.join(l[3:]):60s} {d:8.1f} us")
                continue
            mflop = float(l[-1].split("=")[1])
            key = nm
            if key in seen and not nm.startswith("cond."):
                continue  # print the score layers once (first step)
            seen.add(key)
            print(f"  {nm:28s} {' '.join(l[3:9]):60s} {d:8.1f} us {mflop/d:7.1f} TF/s" if d > 0 else nm)


if __name__ == "__main__":
    main(*sys.argv[1:3])
