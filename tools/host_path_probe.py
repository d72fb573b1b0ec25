"""Host-side time line of one batch-1 enhance call (perf_counter, no profiler): where the host is while the device waits."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import open_universe_amd  # noqa: E402,F401
from open_universe_amd import UniverseGAN  # noqa: E402
from open_universe_amd import _lib  # noqa: E402
from open_universe_amd import config as C  # noqa: E402
from open_universe_amd import state_dict as S  # noqa: E402

spec = C.spec_from_config(C.builtin_config("PP16"))
model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, seed=0), device="cuda:0")
mix = torch.randn(1, 1, 64000, device="cuda:0") * 0.1
rng = torch.Generator(device="cuda:0").manual_seed(1)
marks = []
real_randn, real_call, real_status = torch.randn, model._L.ou_enhance, model._status
n_draw = [0]


def randn(*a, **k):
    if n_draw[0] % 8 == 0:
        marks.append(("first_draw", time.perf_counter()))
    n_draw[0] += 1
    r = real_randn(*a, **k)
    if n_draw[0] % 8 == 0:
        marks.append(("draws_done", time.perf_counter()))
    return r


class L:
    def __getattr__(self, k):
        f = getattr(model_L, k)
        if k != "ou_enhance":
            return f

        def g(*a):
            marks.append(("c_enter", time.perf_counter()))
            r = f(*a)
            marks.append(("c_exit", time.perf_counter()))
            return r
        return g


model_L = model._L
model._L = L()
torch.randn = randn


def status(force=False):
    marks.append(("status_enter", time.perf_counter()))
    real_status(force)
    marks.append(("status_exit", time.perf_counter()))


model._status = status
for _ in range(10):
    model.enhance(mix, rng=rng)
torch.cuda.synchronize()
marks.clear()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    marks.append(("enter", time.perf_counter()))
    model.enhance(mix, rng=rng)
    marks.append(("exit", time.perf_counter()))
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
seq = ["enter", "first_draw", "draws_done", "c_enter", "c_exit", "status_enter", "status_exit", "exit"]
acc = {k: 0.0 for k in seq}
last = None
for k, t in marks:
    if last is not None and k != "enter":
        acc[k] += t - last
    last = t
print(f"{wall * 1e3:.3f} ms per call; host segments (us per call): " + ", ".join(f"->{k} {acc[k] / N * 1e6:.1f}" for k in seq[1:]))
