"""Where does the +-1.5 % spread of the batch-1 figure between bench.py invocations come from?  Several model instances in ONE
process (new handle, side streams, workspace each), several timed loops per instance."""
import gc
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import open_universe_amd  # noqa: E402,F401
from open_universe_amd import UniverseGAN, config as C, state_dict as S  # noqa: E402

spec = C.spec_from_config(C.builtin_config("PP16"))
sd = S.synthetic_state_dict(spec, seed=0)
mix = torch.randn(1, 1, 64000, device="cuda:0") * 0.1
for inst in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    model = UniverseGAN(spec, state_dict=sd, device="cuda:0")
    rng = torch.Generator(device="cuda:0").manual_seed(1)
    for _ in range(10):
        model.enhance(mix, rng=rng)
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            model.enhance(mix, rng=rng)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 60 * 1e3)
    ws = model._ws.data_ptr()
    print(f"instance {inst}: " + " ".join(f"{v:.3f}" for v in res) + f" ms; workspace at {ws:#x}, weights at {model._weights.data_ptr():#x}",
          flush=True)
    del model
    gc.collect()
    torch.cuda.empty_cache()
