"""Equal-length sets: batch size x lanes.   python tools/lanes_batch.py [n_utt]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec, synth_mix
from open_universe_amd import UniverseGAN, distributed as D, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spec = get_spec("PP16")
model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
sigs = [synth_mix(spec, 1, 64000, seed=400 + i)[0].cuda() for i in range(n)]
for bs, k in ((1, 1), (1, 4), (2, 2), (2, 4), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 4), (16, 1), (16, 2)):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        D.enhance_sharded(model, sigs, seed=3, gather=False, batch_size=bs, in_flight=k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"batch_size={bs:2d} in_flight={k}: {1e3 * dt / n:6.2f} ms per utterance, {n / dt:6.1f} utt/s", flush=True)
