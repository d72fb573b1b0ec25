"""Why is the free-running loop ~2 % slower per call than sync-per-call?  (Not the clock: bench.py's sustained leg shows sclk at
2397-2399 MHz throughout.)  Variants: the status copy after every call on / off, side streams inside the call on / off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec, synth_mix
from open_universe_amd import UniverseGAN, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
spec = get_spec("PP16")
model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
mix = synth_mix(spec, 1, 64000).cuda()
rng = torch.Generator(device="cuda").manual_seed(1)
T = 64160
noise = [torch.randn(1, 1, T, device="cuda") for _ in range(8)]


def loop(n, sync, fixed_noise=False, no_status=False):
    model.check_status = sync
    saved = model._status
    if no_status:
        model._status = lambda force=False: None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if fixed_noise:
            model._enhance(mix, 8, None, None, None, None, False, False, None, "median", None, noise)
        else:
            model.enhance(mix, n_steps=8, rng=rng)
        if sync and no_status:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model._status = saved
    model.check_status = True
    return 1e3 * dt / n


loop(5, True)
for tag, kw in (("sync per call", dict(sync=True)), ("free-running", dict(sync=False)),
                ("sync, no status copy", dict(sync=True, no_status=True)), ("free, no status copy", dict(sync=False, no_status=True)),
                ("sync, fixed noise (no randn)", dict(sync=True, fixed_noise=True)), ("free, fixed noise (no randn)", dict(sync=False, fixed_noise=True))):
    print(f"{os.environ.get('OU_NO_OVERLAP', '0')} {tag:32s} {min(loop(60, **kw) for _ in range(3)):.3f} ms per call", flush=True)
