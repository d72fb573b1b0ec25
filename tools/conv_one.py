"""Run ONE packed conv layer repeatedly (ou_bench_conv) -- target for rocprofv3 --pmc.
usage: conv_one.py <layer> <Tin> <iters> [model=PP16] [B=1] [cfg=-1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from ctypes import byref, c_float, c_int32, c_size_t, c_void_p
from helpers import get_spec
from open_universe_amd import Universe, state_dict as S, _lib
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
layer, Tin, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
name = sys.argv[4] if len(sys.argv) > 4 else "PP16"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
cfg = int(sys.argv[6]) if len(sys.argv) > 6 else -1
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
ws = torch.zeros(1 << 31, dtype=torch.uint8, device="cuda")
ms, used = c_float(), c_int32()
_lib.check(model._L.ou_bench_conv(model._handle, layer.encode(), B, Tin, cfg, -1, 1, iters, c_void_p(ws.data_ptr()),
                                  c_size_t(ws.numel()), model._stream(), byref(ms), byref(used)), model._handle)
torch.cuda.synchronize()
print(f"{name} {layer} B={B} cfg{used.value} {ms.value*1e3:.1f} us/launch")
