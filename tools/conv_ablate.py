import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec
from open_universe_amd import Universe, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
spec = get_spec("PP16")
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
p = "_edm_model"
layers = [(p + ".encoder.ds_modules.0.conv1", 64160), (p + ".encoder.ds_modules.1.conv1", 32080),
          (p + ".encoder.ds_modules.3.conv1", 2005), (p + ".encoder.ds_modules.4.conv1", 401)]
for name, Tin in layers:
    out = []
    for dbg, tag in ((0, "full"), (4, "no-epi"), (2, "no-mfma"), (6, "no-mfma,no-epi"), (1, "no-Wld"), (9, "no-ld"), (11, "no-ld,no-mfma"), (15, "nothing")):
        os.environ["OU_DBG"] = str(dbg)
        ms, used = model.bench_conv(name, 1, Tin, with_res=True, iters=10)
        out.append(f"{tag}:{ms*1e3:.1f}")
    print(name[-28:], "cfg", used, " ".join(out))
