"""Tuning aid: conv_direct2_kernel on the deep-level k3 / k5 layers of a model -- K split over 8 or 4 waves x 64 / 32 columns
(force_cfg 105 / 106 / 107 / 108) x XCD ownership (OU_XCD_MAP unset / 1 / 2 / 3 / 4).  Back-to-back launches of one layer
(events around `iters` launches: kernel + dispatch gap, weights warm in the Infinity Cache)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec
from open_universe_amd import Universe, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

name = sys.argv[1] if len(sys.argv) > 1 else "PP16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
maps = [m for m in (sys.argv[3] if len(sys.argv) > 3 else "-1,3,4").split(",")]
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
p = "_edm_model"
T0 = 64160 if name != "PP24" else 96240
layers = [(p + ".encoder.ds_modules.1.conv1", T0 // 2), (p + ".encoder.ds_modules.1.conv2", T0 // 2),
          (p + ".encoder.ds_modules.2.conv1", T0 // 8), (p + ".encoder.ds_modules.2.conv2", T0 // 8),
          (p + ".encoder.ds_modules.3.conv1", T0 // 32), (p + ".encoder.ds_modules.3.conv2", T0 // 32),
          (p + ".encoder.ds_modules.4.conv1", T0 // 160), (p + ".encoder.ds_modules.4.conv2", T0 // 160)]
if name == "PP24":
    layers = [(n, t) for n, t in layers[4:]] + [(p + ".encoder.ds_modules.4.conv1", T0 // 240), (p + ".encoder.ds_modules.4.conv2", T0 // 240)]
    layers = [(p + ".encoder.ds_modules.3.conv1", T0 // 40), (p + ".encoder.ds_modules.3.conv2", T0 // 40),
              (p + ".encoder.ds_modules.4.conv1", T0 // 240), (p + ".encoder.ds_modules.4.conv2", T0 // 240)]
for lname, Tin in layers:
    for mp in maps:
        if mp == "-1":
            os.environ.pop("OU_XCD_MAP", None)
        else:
            os.environ["OU_XCD_MAP"] = mp
        row = []
        for tag, cfg in (("auto", -1), ("wk8.tn2", 105), ("wk8.tn1", 106), ("wk4.tn2", 107), ("wk4.tn1", 108), ("wino8", 109), ("wino4", 110), ("d4w.tm1", 610), ("d4w.tm2", 620), ("d4w4.tm1", 710), ("d4w4.tm2", 720)):
            if cfg >= 600: cfg += 5 if 'conv1' in lname else 3
            try:
                best = None
                for rep in range(3):
                    ms, used = model.bench_conv(lname, B, Tin, cfg=cfg, with_res=True, iters=30)
                    best = ms if best is None else min(best, ms)
                row.append(f"{tag}:{best*1e3:6.1f}us(cfg{used})")
            except Exception as e:
                row.append(f"{tag}: n/a")
        print(f"{lname[-36:]:36s} T={Tin:6d} map={mp:>2s} " + "  ".join(row), flush=True)
os.environ.pop("OU_XCD_MAP", None)
