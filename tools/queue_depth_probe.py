"""Free-running calls with a bounded number of calls enqueued ahead of the device (depth d: the host waits for the end of call
i - d before it enqueues call i + 1).  d = 0 is the product default (status wait per call), inf = free-running."""
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
for _ in range(200):
    model.enhance(mix, rng=rng)


def loop(depth, n=80):
    model.check_status = depth == 0
    evs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        model.enhance(mix, rng=rng)
        if depth and depth < 10 ** 6:
            e = torch.cuda.Event()
            e.record()
            evs.append(e)
            if len(evs) > depth:
                evs.pop(0).synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    model.check_status = True
    model._status(force=True)
    return dt


for rep in range(3):
    print("  ".join(f"d={('inf' if d >= 10 ** 6 else d)}: {loop(d):.3f}" for d in (0, 1, 2, 4, 10 ** 6)), flush=True)
