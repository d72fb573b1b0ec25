"""Host cost of the library's dry walks (layout of a call is computed by walking the network without launching)."""
import sys
import time

sys.path.insert(0, ".")
from ctypes import byref, c_size_t  # noqa: E402

import open_universe_amd  # noqa: E402,F401
from open_universe_amd import UniverseGAN, config as C, state_dict as S  # noqa: E402

spec = C.spec_from_config(C.builtin_config("PP16"))
model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, seed=0), device="cuda:0")
c = c_size_t()
for _ in range(10):
    model._L.ou_workspace_bytes(model._handle, 1, 64160, byref(c))
t0 = time.perf_counter()
for _ in range(200):
    model._L.ou_workspace_bytes(model._handle, 1, 64160, byref(c))
print(f"ou_workspace_bytes (dry walk of conditioner + one score pass): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
