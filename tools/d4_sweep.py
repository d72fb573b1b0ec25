"""Tuning aid for conv_direct4_kernel (1x1 / phase-GEMM / rate-change layers): every tile shape (TM = 2 / 4 rows x 16, WK = 1 / 2 /
4 / 8 slices of the reduction) against the launcher's own choice and against the first-generation kernels (OU_CONV_DIRECT=3)
on every layer of a model the family can take.   python tools/d4_sweep.py [PP16|OR16|PP24] [B]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402,F401
from helpers import get_spec  # noqa: E402
from open_universe_amd import Universe, state_dict as S  # noqa: E402
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

name = sys.argv[1] if len(sys.argv) > 1 else "PP16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
layers = json.loads(model._L.ou_plan_json(model._handle).decode())["convs"]
T0 = int(4 * spec.fs) + (spec.tot_ds - int(4 * spec.fs) % spec.tot_ds)
C0 = spec.score.n_channels
lvlT, t, c = {}, T0, C0
for r in list(spec.score.rate_factors) + [1]:
    lvlT[c] = t
    t //= r
    c *= 2
seen = set()
tot_auto = tot_old = 0.0
for L in layers:
    if L["name"] in seen:
        continue
    seen.add(L["name"])
    R = L["stride"] if L["stride"] > 1 else 1
    if L["KW"] != R or L["pad"] != 0 or L["Cin"] % 16:
        continue
    Tin = lvlT.get(L["Cin"], T0 // spec.tot_ds)
    Nq = Tin // R
    mflop = 2.0 * L["M"] * Nq * L["Cin"] * L["KW"] * B * 1e-6
    row = []
    best = None
    for tm in (1, 2, 3, 4):
        for lw, wk in enumerate((1, 2, 4, 8)):
            if tm in (1, 3) and wk < 4:
                continue
            cfg = 300 + 10 * tm + lw
            try:
                ms, used = model.bench_conv(L["name"], B, Tin, cfg=cfg, with_res=False, iters=20)
                row.append(f"t{tm}k{wk}:{ms * 1e3:5.1f}")
                if best is None or ms < best[0]:
                    best = (ms, f"t{tm}k{wk}")
            except Exception:
                row.append(f"t{tm}k{wk}:  n/a")
    auto, used = model.bench_conv(L["name"], B, Tin, with_res=False, iters=20)
    os.environ["OU_D4_SHORT"] = "1"
    short, used_s = model.bench_conv(L["name"], B, Tin, with_res=False, iters=20)
    os.environ.pop("OU_D4_SHORT")
    os.environ["OU_CONV_DIRECT"] = "3"
    old, used_old = model.bench_conv(L["name"], B, Tin, with_res=False, iters=20)
    os.environ.pop("OU_CONV_DIRECT")
    tot_auto += auto
    tot_old += old
    bs = f"best {best[1]} {best[0] * 1e3:5.1f}" if best else "best n/a"
    print(f"{L['name'][-44:]:44s} M={L['M']:5d} Nq={Nq:6d} K={L['Cin'] * L['KW']:5d} up={L['up']} | auto cfg{used} {auto * 1e3:5.1f}us "
          f"{mflop / auto / 1e6:5.1f}TF | short-rule cfg{used_s} {short * 1e3:5.1f}us | {bs} | old cfg{used_old} {old * 1e3:5.1f}us {mflop / old / 1e6:5.1f}TF | " + " ".join(row), flush=True)
print(f"sum over the layers: auto {tot_auto * 1e3:.1f} us, first generation {tot_old * 1e3:.1f} us")
