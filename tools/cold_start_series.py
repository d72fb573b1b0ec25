"""Per-20-call averages from a fresh process on: model creation, 5 warm-up calls, then back-to-back timed loops (what bench.py's
headline loop sees first, and what it would see later)."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import open_universe_amd  # noqa: E402,F401
from open_universe_amd import UniverseGAN, config as C, state_dict as S  # noqa: E402

spec = C.spec_from_config(C.builtin_config("PP16"))
model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, seed=0), device="cuda:0")
mix = torch.randn(1, 1, 64000, device="cuda:0") * 0.1
rng = torch.Generator(device="cuda:0").manual_seed(1)
for _ in range(5):
    model.enhance(mix, rng=rng)
out = []
t_start = time.perf_counter()
for _ in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        model.enhance(mix, rng=rng)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 20 * 1e3)
print(" ".join(f"{v:.2f}" for v in out), f"| {time.perf_counter() - t_start:.1f} s")
