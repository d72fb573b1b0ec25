"""(needs the experiments library: make -C open-universe_amd/csrc EXPERIMENTS=1 and OU_LIBRARY=.../lib/libouniverse_experiments.so --
the default library ignores the switches that invalidate results)
Upper bound for "the first decoder block under the GRU pass" (review of round 3, item 3): time of a score forward with the
three convs of score.dec0 launched on a side stream that does NOT wait for the recurrence (OU_DBG_DEC0=1: results invalid), against
the normal forward.  A scheme that gates those convs on the recurrence's progress can only be slower than this."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec
from open_universe_amd import UniverseGAN, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
spec = get_spec("PP16")
model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
T = 64160
xin = torch.randn(1, 1, T, device="cuda")
model.condition_model(xin)
sig = torch.full((1,), 0.5)


def fwd_ms(K=40):
    for _ in range(3):
        model.score_model(xin, sig)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        model.score_model(xin, sig)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


for rep in range(3):
    os.environ.pop("OU_DBG_DEC0", None)
    a = fwd_ms()
    os.environ["OU_DBG_DEC0"] = "1"
    b = fwd_ms()
    model.profile(True)
    model.score_model(xin, sig)
    torch.cuda.synchronize()
    recs = model.profile_read()
    model.profile(False)
    gru = [r[0] * 1e3 for r in recs if r[3] >= 1000]
    os.environ.pop("OU_DBG_DEC0")
    model.profile(True)
    model.score_model(xin, sig)
    torch.cuda.synchronize()
    recs0 = model.profile_read()
    model.profile(False)
    gru0 = [r[0] * 1e3 for r in recs0 if r[3] >= 1000]
    print(f"score forward: normal {a * 1e3:.1f} us, dec0 under the GRU pass (invalid results) {b * 1e3:.1f} us, difference {1e3 * (a - b):+.1f} us; "
          f"GRU pass alone {gru0[0]:.1f} us, with the three convs beside it {gru[0]:.1f} us", flush=True)
