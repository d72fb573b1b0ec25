"""Experiment: two enhance calls in flight on two streams (two handles sharing one packed weight blob, one workspace each)
against one call of twice the batch -- does the GRU pass of one call hide under the convs of the other?
usage: dual_stream.py [model=PP16] [B=4] [iters=10]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec, synth_mix
from open_universe_amd import Universe, _lib, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

name = sys.argv[1] if len(sys.argv) > 1 else "PP16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
nstream = int(sys.argv[4]) if len(sys.argv) > 4 else 2
spec = get_spec(name)
blob, _ = _lib.pack_weights(spec, S.synthetic_state_dict(spec, 0))
blob = blob.cuda()
models = [Universe(spec, packed_weights=blob, device="cuda:0") for _ in range(nstream)]
for m in models:
    m.check_status = False
T = 4 * spec.fs
mixes = [synth_mix(spec, B, T, seed=1000 + 50 * i).cuda() for i in range(nstream)]
big = torch.cat(mixes, dim=0)
streams = [torch.cuda.Stream() for _ in range(nstream)]
rngs = [torch.Generator(device="cuda").manual_seed(7 + i) for i in range(nstream)]


def run_dual():
    for m, x, s, g in zip(models, mixes, streams, rngs):
        with torch.cuda.stream(s):
            m.enhance(x, n_steps=8, rng=g)


def run_big():
    models[0].enhance(big, n_steps=8, rng=rngs[0])


def run_seq():
    for x in mixes:
        models[0].enhance(x, n_steps=8, rng=rngs[0])


for fn in (run_dual, run_big, run_seq):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name} {fn.__name__:9s}: {nstream} x B={B}: {dt*1e3:8.2f} ms per round -> {nstream*B/dt:7.1f} utt/s", flush=True)
for m in models:
    m.synchronize()
