"""Fused ConvBlock body vs the three generic launches: same model, same inputs, every fuse mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec, synth_mix
from open_universe_amd import Universe, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import restatement as O

name = sys.argv[1] if len(sys.argv) > 1 else "PP16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 40123
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
mix = synth_mix(spec, B, T).cuda()
Tp = T + (spec.tot_ds - T % spec.tot_ds)
noise = [z.cuda() for z in torch.randn(4, B, 1, Tp, generator=torch.Generator().manual_seed(5))]
outs = {}
for mode, nc in (("0", ""), ("3", ""), ("2", ""), ("3", "128"), ("3", "256"), ("2", "128"), ("2", "256"), ("-1", "")):
    os.environ["OU_FUSE"] = mode
    if nc: os.environ["OU_FUSE_NC"] = nc
    else: os.environ.pop("OU_FUSE_NC", None)
    if getattr(model, '_ws', None) is not None: model._ws.zero_()
    y = model._enhance(mix, 4, None, None, None, None, False, False, None, 'median', None, noise)
    torch.cuda.synchronize()
    outs[(mode, nc)] = y.cpu()
    ref = outs[("0", "")]
    vs = {k: model.tensor(k).cpu() for k in ("score.enc0.v", "score.enc1.v", "score.dec3.v", "score.dec4.v", "cond.enc0.v", "cond.dec4.v")}
    if mode == "0": vref = vs
    print("   |y|", float(y.abs().max()), "maxdiff", float((y.cpu() - ref).abs().max()),
          {k: f"{O.si_sdr(vref[k].flatten(), v.flatten()):.0f}" for k, v in vs.items()})
    print(f"OU_FUSE={mode:>2} NC={nc or 'auto':>4}: SI-SDR vs unfused {min(O.si_sdr(ref[b], y.cpu()[b]) for b in range(B)):.1f} dB"
          f"  launches {model.launch_stats()}")
