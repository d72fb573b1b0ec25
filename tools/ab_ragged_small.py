"""Rate of small ragged batches (2 / 3 utterances of different lengths per call) -- run once per library (OU_LIBRARY)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import open_universe_amd  # noqa: E402,F401
from open_universe_amd import UniverseGAN as Universe  # noqa: E402
from open_universe_amd import config as C  # noqa: E402
from open_universe_amd import state_dict as S  # noqa: E402

spec = C.spec_from_config(C.builtin_config("PP16"))
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, seed=0), device="cuda:0")
g = torch.Generator().manual_seed(5)
for B in (2, 3):
    lens = [int(spec.fs * (3.5 + 0.4 * float(torch.rand(1, generator=g)))) for _ in range(B)]
    sigs = [torch.randn(L, generator=g).cuda() * 0.1 for L in lens]
    rng = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(3):
        model.enhance_many(sigs, rngs=rng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        model.enhance_many(sigs, rngs=rng)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"ragged B={B}: {dt * 1e3:.2f} ms per call, {B / dt:.1f} utt/s, launches {sum(model.launch_stats())}")
