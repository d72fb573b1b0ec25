"""Where a batch-1 enhance call's wall time goes that is NOT kernels: the bubble between two calls (host path of the wrapper:
status wait, Python, first launch) and the idle time inside a call, from a rocprofv3 kernel trace.

  run:      rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cb -- python tools/call_boundary.py run
  analyse:  python tools/call_boundary.py analyse gpurun_out/cb
"""
import csv
import glob
import sys

sys.path.insert(0, ".")


def run():
    import torch

    import open_universe_amd  # noqa: F401
    from open_universe_amd import UniverseGAN
    from open_universe_amd import config as C
    from open_universe_amd import state_dict as S

    spec = C.spec_from_config(C.builtin_config("PP16"))
    model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, seed=0), device="cuda:0")
    mix = torch.randn(1, 1, 64000, device="cuda:0") * 0.1
    rng = torch.Generator(device="cuda:0").manual_seed(1)
    model.check_status = "free" not in sys.argv  # (`run free`: no status wait per call)
    for _ in range(40):
        model.enhance(mix, rng=rng)
    torch.cuda.synchronize()


def analyse(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # calls: from the first kernel behind a post_reg_kernel to the next post_reg_kernel
    calls, cur = [], []
    for s, e, n in rows:
        cur.append((s, e, n))
        if "post_reg_kernel" in n:
            calls.append(cur)
            cur = []
    calls = [c for c in calls if any("pad_normalize" in n for _, _, n in c)][10:]  # (warm calls only)
    bub, draws, idle, wall, busy = [], [], [], [], []
    for i, c in enumerate(calls):
        t0, t1 = c[0][0], c[-1][1]
        wall.append(t1 - t0)
        tp = [s for s, e, n in c if "pad_normalize" in n][0]
        draws.append(tp - t0)
        # union of the intervals
        u, end = 0, t0
        for s, e, n in c:
            if e > end:
                u += e - max(s, end)
                end = e
        busy.append(u)
        idle.append((t1 - t0) - u)
        if i:
            bub.append(t0 - calls[i - 1][-1][1])
    m = lambda v: sum(v) / max(1, len(v)) / 1e3  # noqa: E731
    print(f"{len(calls)} warm calls: kernels of a call span {m(wall):.1f} us (device busy {m(busy):.1f}, idle inside the call "
          f"{m(idle):.1f} us over {sum(len(c) for c in calls) / len(calls):.0f} kernels); draws + their gaps in front of the "
          f"first library kernel {m(draws):.1f} us; bubble between the last kernel of a call and the first of the next "
          f"{m(bub):.1f} us  ->  period {m(wall) + m(bub):.1f} us")
    # the largest idle gaps inside a call, by the kernel that follows them
    gaps = {}
    for c in calls:
        end = c[0][1]
        for (s, e, n), (ps, pe, pn) in zip(c[1:], c[:-1]):
            g = s - end
            if g > 0:
                k = (pn.split("(")[0][-40:], n.split("(")[0][-40:])
                gaps.setdefault(k, []).append(g)
            end = max(end, e)
    top = sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]
    for (a, b), v in top:
        print(f"  {sum(v) / len(calls) / 1e3:7.1f} us per call in {len(v) / len(calls):5.1f} gaps of {sum(v) / len(v) / 1e3:5.1f} us: {a} -> {b}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        analyse(sys.argv[2])
