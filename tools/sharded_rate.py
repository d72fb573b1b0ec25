"""Utterances per second through the PRODUCT sharded path (open_universe_amd.distributed.enhance_sharded) on one GPU: one
`enhance` call per utterance (batch_size=1, what round 2 did) against groups of 4 / 8 equal-length utterances per call.
usage: sharded_rate.py [model=PP16] [n_utterances=32]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec, synth_mix
from open_universe_amd import Universe, distributed as D, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

name = sys.argv[1] if len(sys.argv) > 1 else "PP16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
sigs = [synth_mix(spec, 1, 4 * spec.fs, seed=2000 + i)[0].cuda() for i in range(n)]
for bs in (1, 4, 8):
    D.enhance_sharded(model, sigs[:bs], seed=1, gather=False, batch_size=bs, n_steps=8)  # warm-up (workspace, kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = D.enhance_sharded(model, sigs, seed=1, gather=False, batch_size=bs, n_steps=8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert len(out) == n and all(torch.isfinite(o).all() for o in out.values())
    print(f"{name}: enhance_sharded({n} utterances of 4 s, 8 steps, batch_size={bs}) on one GPU: {dt*1e3:8.1f} ms -> "
          f"{n/dt:6.1f} utterances/s, RTF {4*n/dt:7.1f}x", flush=True)
