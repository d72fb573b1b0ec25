"""What runs beside what with K lanes: every conv / GRU launch of every lane stamps its own start and end on the device's
constant clock (ou_profile_*); this lays the launches of all lanes on one timeline.  (rocprofv3 --kernel-trace serialises the
streams of a process, so its trace shows ONE kernel at a time whatever the lanes do.)   python tools/lanes_timeline.py K n_utt"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from helpers import get_spec, synth_mix  # noqa: E402
from open_universe_amd import UniverseGAN, distributed as D, state_dict as S  # noqa: E402
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
from open_universe_amd.lanes import LanePool  # noqa: E402

K, n = int(sys.argv[1]), int(sys.argv[2])
spec = get_spec("PP16")
model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
sigs = [synth_mix(spec, 1, 64000 - 37 * i, seed=400 + i)[0].cuda() for i in range(n)]
D.enhance_sharded(model, sigs, seed=3, gather=False, in_flight=K)  # lanes, workspaces
models = [model] + model.__dict__.get("_lane_forks", [])[:K - 1]
for m in models:
    m.profile(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
D.enhance_sharded(model, sigs, seed=3, gather=False, in_flight=K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"in_flight={K}: {1e3 * dt / n:.2f} ms per utterance, {n / dt:.1f} utt/s (per-launch stamps on)")


def kind(c):
    if c >= 1000: return "gru"
    if c in (66, 76): return "direct2 (k3/k5 deep)"
    if 100 <= c < 200: return "chain (C=32/64 bodies)"
    if 300 <= c < 400: return "direct4 (1x1/rate)"
    if 50 <= c < 100: return "direct gen1 (1x1/rate)"
    if 40 <= c < 50: return "rate_up/down"
    return "other conv"


ev = []
for li, m in enumerate(models):
    for a, b, c in m.profile_read_ticks():
        if b > a:
            ev.append((a, b, li, kind(c)))
    m.profile(False)
lo, hi = min(e[0] for e in ev), max(e[1] for e in ev)
a0, b0 = lo + 0.15 * (hi - lo), lo + 0.9 * (hi - lo)
ev = [e for e in ev if e[0] >= a0 and e[1] <= b0]
pts = []
for s, e, li, k in ev:
    pts.append((s, 1, k))
    pts.append((e, -1, k))
pts.sort()
depth, last, hist = 0, a0, collections.Counter()
gru_on, with_gru = 0, collections.Counter()
for t, d, k in pts:
    hist[depth] += t - last
    with_gru[(gru_on > 0, depth - gru_on)] += t - last
    last = t
    depth += d
    if k == "gru":
        gru_on += d
tot = sum(hist.values())
print(f"window {(b0 - a0) / 1e5:.1f} ms, {len(ev)} profiled launches (convs + GRU passes; FIR / in / out conv and glue are not stamped)")
print("profiled launches in flight: " + "  ".join(f"{k}: {100 * v / tot:.1f}%" for k, v in sorted(hist.items())))
print("GRU passes x conv launches in flight: " + "  ".join(f"gru={'y' if g else 'n'},convs={c}: {100 * v / tot:.1f}%" for (g, c), v in sorted(with_gru.items())))
agg = collections.defaultdict(list)
for s, e, li, k in ev:
    agg[k].append((e - s) / 100.0)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:26s} launches {len(v):5d}  avg {sum(v) / len(v):7.1f} us  total {sum(v) / 1e3:7.2f} ms  ({sum(v) / 1e3 / (len(ev) and n * 0.75):.2f} ms per utterance)")
