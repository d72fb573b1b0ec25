import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec, synth_mix
from open_universe_amd import Universe, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
name, B, bmax, frames = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
if bmax != "0":
    os.environ["OU_GRU_BMAX"] = bmax
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
model.check_status = False
mix = synth_mix(spec, B, spec.tot_ds * frames).cuda()
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 2
fails = 0
import time
for it in range(iters):
    t0 = time.time()
    out = model.enhance(mix, n_steps=2, rng=torch.Generator(device="cuda").manual_seed(0))
    torch.cuda.synchronize()
    hdr = model._ws[:256].view(torch.int32).cpu().tolist()
    fails += int(hdr[0] != 0)
    if hdr[0] != 0 or it == iters - 1: print(f"fails {fails}/{it+1} last call {time.time()-t0:.3f}s",name, "B", B, "bmax", bmax, "frames", frames, "overlap", os.environ.get("OU_NO_OVERLAP") is None, "iter", it, "status", hdr[0], "epochs", hdr[2:6], "rdv", hdr[8:11], "gather[cluster,g,step,min,want]", hdr[12:17], "xcc,plain,bid", hdr[17:20],  "env", {k: v for k, v in os.environ.items() if k.startswith("OU_")}, flush=True)
    model._ws[:4].zero_()
    model._status_host.zero_()
