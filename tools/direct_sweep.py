"""Tuning aid: the direct conv kernels with 32- vs 64-column tiles (force_cfg 106 / 105) against the LDS kernel (cfg -1
with OU_CONV_DIRECT=0) on the deep-level layer shapes of PP16."""
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
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
p = "_edm_model"
T0 = 64160
layers = [(p + ".encoder.ds_modules.1.conv1", T0 // 2), (p + ".encoder.ds_modules.1.rate_change_conv", T0 // 2),
          (p + ".encoder.ds_modules.2.conv1", T0 // 8), (p + ".encoder.ds_modules.2.conv2", T0 // 8),
          (p + ".encoder.ds_modules.2.rate_change_conv", T0 // 8),
          (p + ".encoder.ds_modules.3.conv1", T0 // 32), (p + ".encoder.ds_modules.3.conv2", T0 // 32),
          (p + ".encoder.ds_modules.3.rate_change_conv", T0 // 32),
          (p + ".encoder.ds_modules.4.conv1", T0 // 160), (p + ".encoder.ds_modules.4.conv2", T0 // 160),
          (p + ".encoder.gru#l0", T0 // 160), (p + ".decoder.up_modules.1.rate_change_conv", T0 // 160),
          (p + ".decoder.up_modules.2.rate_change_conv", T0 // 32), (p + ".decoder.up_modules.3.rate_change_conv", T0 // 8),
          (p + ".decoder.up_modules.4.rate_change_conv", T0 // 2), ("condition_model.encoder.st_convs.0", T0 // 160)]
for lname, Tin in layers:
    row = []
    for tag, cfg, env in (("auto", -1, None), ("tn1", 106, None), ("tn2", 105, None), ("lds", -1, "0")):
        if env is not None:
            os.environ["OU_CONV_DIRECT"] = env
        try:
            ms, used = model.bench_conv(lname, B, Tin, cfg=cfg, with_res=False, iters=20)
            row.append(f"{tag}:{ms*1e3:6.1f}us(cfg{used})")
        except Exception as e:
            row.append(f"{tag}: n/a")
        os.environ.pop("OU_CONV_DIRECT", None)
    print(f"{lname[-42:]:42s} T={Tin:6d} " + "  ".join(row), flush=True)
