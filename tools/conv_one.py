"""Run ONE packed conv layer of PP16 repeatedly (ou_bench_conv) -- target for rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from ctypes import byref, c_float, c_int32, c_size_t, c_void_p
from helpers import get_spec
from open_universe_amd import Universe, state_dict as S, _lib
layer, Tin, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
spec = get_spec("PP16")
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
ws = torch.zeros(1 << 29, dtype=torch.uint8, device="cuda")
ms, used = c_float(), c_int32()
_lib.check(model._L.ou_bench_conv(model._handle, layer.encode(), 1, Tin, -1, -1, 1, iters, c_void_p(ws.data_ptr()),
                                  c_size_t(ws.numel()), model._stream(), byref(ms), byref(used)), model._handle)
torch.cuda.synchronize()
print(f"{layer} cfg{used.value} {ms.value*1e3:.1f} us/launch")
