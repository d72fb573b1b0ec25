"""Tuning aid: sweep tile configs / chunks-per-stage of the generic conv kernel over the PP16 layer shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec
from open_universe_amd import Universe, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

spec = get_spec(sys.argv[1] if len(sys.argv) > 1 else "PP16")
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = "_edm_model"
only = sys.argv[3] if len(sys.argv) > 3 else ""
layers = [  # (prefix, Tin, MFLOP per batch element (reference accounting))
    (p + ".encoder.ds_modules.0.conv1", 64160, 657), (p + ".encoder.ds_modules.0.conv2", 64160, 394),
    (p + ".encoder.ds_modules.0.rate_change_conv", 64160, 263),
    (p + ".encoder.ds_modules.1.conv1", 32080, 1314), (p + ".encoder.ds_modules.1.rate_change_conv", 32080, 526),
    (p + ".encoder.ds_modules.2.conv1", 8020, 1314), (p + ".encoder.ds_modules.2.rate_change_conv", 8020, 526),
    (p + ".encoder.ds_modules.3.conv1", 2005, 1314), (p + ".encoder.ds_modules.3.rate_change_conv", 2005, 526),
    (p + ".encoder.ds_modules.4.conv1", 401, 1051), (p + ".encoder.ds_modules.4.conv2", 401, 631),
    (p + ".encoder.gru#l0", 401, 630),
    (p + ".decoder.up_modules.1.rate_change_conv", 401, 526), (p + ".decoder.up_modules.4.rate_change_conv", 32080, 263),
    ("condition_model.encoder.st_convs.0", 401, 2102),
]
for name, Tin, mflop in layers:
    if only and only not in name:
        continue
    res = []
    for cfg in range(8):
        for sc in (1, 2, 4):
            try:
                ms, used = model.bench_conv(name, B, Tin, cfg=cfg, sc=sc, with_res=True, iters=10)
                res.append((ms, cfg, sc))
            except Exception as e:
                pass
    auto, used = model.bench_conv(name, B, Tin, with_res=True, iters=10)
    res.sort()
    best = res[0]
    tf = lambda ms: mflop * B / ms / 1e6 * 1e3 / 1e3
    print(f"{name[-40:]:40s} T={Tin:6d} auto cfg{used} {auto*1e3:7.1f}us {tf(auto):6.1f}TF | best cfg{best[1]} sc{best[2]} {best[0]*1e3:7.1f}us {tf(best[0]):6.1f}TF | "
          + " ".join(f"c{c}s{s}:{m*1e3:.0f}" for m, c, s in sorted(res, key=lambda r: (r[1], r[2]))))
