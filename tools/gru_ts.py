import os, sys
os.environ["OU_GRU_TS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec, synth_mix
from open_universe_amd import Universe, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
spec = get_spec("PP16")
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
mix = synth_mix(spec, 1, 64000).cuda()
for _ in range(2):
    model.enhance(mix, n_steps=2, rng=torch.Generator(device="cuda").manual_seed(0))
torch.cuda.synchronize()
ws = model._ws
ts = ws[ws.numel() - (1 << 20):].view(torch.int64)[: 64 * 8 * 8].view(64, 8, 8).cpu().double()
v = os.environ.get("OU_GRU_V", "2")
for blk in (0, 1, 8, 9):
    if v == "1":
        print("v1 block", blk, "per-step cycles [compute, poll(wave0)/idle, barrier | matvec, reduce, gates+stores]:",
              [[round(float(x) / 401) for x in ts[blk, w, [0, 1, 2, 4, 5, 6]]] for w in (0, 1, 7)])
    else:
        print("v2 block", blk, "per-step cycles per wave [compute (matvec..publish), gather (poll until all tags)]:",
              [[round(float(x) / 401) for x in ts[blk, w, [0, 1]]] for w in (0, 1, 2, 3)],
              "| us: entry -> first step", [round(float(ts[blk, w, 4]) / 100, 1) for w in (0, 1, 2, 3)],
              "steps", [round(float(ts[blk, w, 5]) / 100, 1) for w in (0, 1, 2, 3)],
              "| poll rounds per step", [round(float(ts[blk, w, 2]) / 400, 2) for w in (0, 1, 2, 3)],
              "| store ack (OU_GRU_BACKOFF=10)", [round(float(ts[blk, w, 7]) / 401) for w in (0, 1, 2, 3)])
if v != "1":
    st = ts[:, :, 6]
    on = st > 0
    print("kernel entry spread over the participating waves: %.1f us" % ((st[on].max() - st[on].min()).item() / 100))
