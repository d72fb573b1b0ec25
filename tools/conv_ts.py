import os, sys
os.environ["OU_TS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, ctypes
from ctypes import byref, c_float, c_int32, c_size_t, c_void_p
from helpers import get_spec
from open_universe_amd import Universe, state_dict as S, _lib
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
spec = get_spec("PP16")
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
p = "_edm_model"
layers = [(p + ".encoder.ds_modules.0.conv1", 64160, 502, 4), (p + ".encoder.ds_modules.1.conv1", 32080, 502, 4),
          (p + ".encoder.ds_modules.2.conv1", 8020, 504, 8),
          (p + ".encoder.ds_modules.3.conv1", 2005, 256, 8), (p + ".encoder.ds_modules.4.conv1", 401, 208, 8),
          (p + ".encoder.ds_modules.4.conv2", 401, 208, 8)]
ws = torch.zeros(1 << 29, dtype=torch.uint8, device="cuda")
for name, Tin, nblk, nw in layers:
    ms, used = c_float(), c_int32()
    _lib.check(model._L.ou_bench_conv(model._handle, name.encode(), 1, Tin, -1, -1, 1, 3, c_void_p(ws.data_ptr()),
                                      c_size_t(ws.numel()), model._stream(), byref(ms), byref(used)), model._handle)
    torch.cuda.synchronize()
    ts = ws[ws.numel() - (16 << 20):].view(torch.int64)[: nblk * nw * 8].view(nblk, nw, 8).cpu().double(); ws[ws.numel() - (16 << 20):].zero_()
    t0 = ts[..., 7]
    start_spread = (t0.max() - t0.min()).item()
    m = ts.mean(dim=(0, 1))
    tot = m[:5].sum().item()
    print(f"{name[-26:]} cfg{used.value} {ms.value*1e3:.1f}us | cycles/wave: goff+ld0 {m[0]:.0f} st0+bar {m[1]:.0f} mainloop {m[2]:.0f} (mma {m[5]:.0f} wait {m[6]:.0f}) epi-lds {m[3]:.0f} epi-out {m[4]:.0f} total {tot:.0f} | start spread {start_spread:.0f}")
