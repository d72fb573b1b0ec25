"""One multi-lane run for tracing: python tools/lanes_run.py K n_utt [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec, synth_mix
from open_universe_amd import UniverseGAN, distributed as D, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
K, n = int(sys.argv[1]), int(sys.argv[2])
spec = get_spec("PP16")
model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
sigs = [synth_mix(spec, 1, 64000 - 37 * i, seed=400 + i)[0].cuda() for i in range(n)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    D.enhance_sharded(model, sigs, seed=3, gather=False, in_flight=K)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"in_flight={K} rep {rep}: {1e3 * dt / n:.2f} ms per utterance, {n / dt:.1f} utt/s", flush=True)
