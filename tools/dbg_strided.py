import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import restatement as O
from helpers import get_spec, synth_mix
from open_universe_amd import Universe, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
spec = get_spec("PP16")
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
mix = synth_mix(spec, 1, 12160 - 160)
xin = O.normalize(torch.nn.functional.pad(mix[:, None, :], (80, 80)), spec.level_db).cuda()
outs = {}
for d in ("0", "1", "1w", "1s"):
    os.environ["OU_CONV_DIRECT"] = d[0]
    os.environ["OU_DBG"] = {"1w": "16", "1s": "8"}.get(d, "0")
    model.condition_model(xin)
    torch.cuda.synchronize()
    outs[d] = {n: model.tensor(n).clone() for n in ("cond.enc1.h", "cond.enc2.h", "cond.enc3.h", "cond.enc1.v")}
for n, d in [(n, d) for d in ("1", "1w", "1s") for n in outs["0"]]:
    a, b = outs["0"][n][0].cpu(), outs[d][n][0].cpu()
    print("variant", d, end=" ")
    bad = ~torch.isfinite(b)
    print(n, tuple(a.shape), "nan/inf:", int(bad.sum()), "of", b.numel(), "si-sdr(finite part):",
          O.si_sdr(a[~bad], b[~bad]) if (~bad).any() else None)
    if bad.any():
        rows = bad.any(dim=1).nonzero().flatten().tolist()
        cols = bad.any(dim=0).nonzero().flatten().tolist()
        print("   bad rows:", rows[:12], "...", len(rows), " bad cols:", cols[:12], "...", len(cols))
    else:
        d_ = (a - b).abs()
        print("   max abs diff", float(d_.max()), "at", divmod(int(d_.argmax()), a.shape[1]), "ref rms", float(a.std()))
        # error by 32-row block / column region
        blk = d_.view(a.shape[0] // 32, 32, -1).amax(dim=(1,))
        print("   per-row-block max:", [round(float(v), 4) for v in blk.amax(dim=1)][:8], "first cols:", [round(float(v), 4) for v in d_.amax(dim=0)[:8]], "last cols:", [round(float(v), 4) for v in d_.amax(dim=0)[-8:]])
