"""Tuning aid: where the waves of conv_direct3w_kernel (the no-split-K minimal-filtering kernel of the throughput regime) spend
their loop: cycles in the loop, cycles inside its s_waitcnt vmcnt (waiting for operands), MFMA cycles of the wave (OU_TS stamps).
  python tools/d3_ts.py [PP16|PP24] [B]"""
import os, sys
os.environ["OU_TS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import json
import torch
from ctypes import byref, c_float, c_int32, c_size_t, c_void_p
from helpers import get_spec
from open_universe_amd import Universe, state_dict as S, _lib
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

name = sys.argv[1] if len(sys.argv) > 1 else "PP16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
plan = {c["name"]: c for c in json.loads(model._L.ou_plan_json(model._handle).decode())["convs"]}
pfx = spec.score_prefix
T = (4 * spec.fs // spec.tot_ds + 1) * spec.tot_ds
ws = torch.zeros(3 << 29, dtype=torch.uint8, device="cuda")
tail = ws[ws.numel() - (16 << 20):].view(torch.int64)
for i, r in enumerate([1] + list(spec.score.rate_factors)):
    T //= r
    for cv in ("conv1", "conv2"):
        lname = f"{pfx}.encoder.ds_modules.{i}.{cv}"
        L = plan[lname]
        ms, used = c_float(), c_int32()
        tail.zero_()
        _lib.check(model._L.ou_bench_conv(model._handle, lname.encode(), B, T, -1, -1, 1, 3, c_void_p(ws.data_ptr()),
                                          c_size_t(ws.numel()), model._stream(), byref(ms), byref(used)), model._handle)
        torch.cuda.synchronize()
        u = used.value
        if not 500 <= u < 600:
            print(f"{lname[-28:]:28s} cfg{u} {ms.value * 1e3:7.1f} us (not conv_direct3w_kernel)")
            continue
        tm, kw = (u % 100) // 10, u % 10
        ts = tail[: 2048 * 4 * 8].view(-1, 8).cpu().double()
        ts = ts[ts[:, 7] > 0]
        m, mx = ts.mean(dim=0), ts.max(dim=0).values
        mfma = L["Cin"] / 4 * 2 * (kw + 1) * tm * 32
        flop = 2.0 * L["M"] * L["Cin"] * L["KW"] * T * B
        print(f"{lname[-28:]:28s} cfg{u} C={L['Cin']:4d} T={T:6d} B={B} {ms.value * 1e3:7.1f} us {flop / ms.value / 1e9:6.1f} TF/s alg | {ts.shape[0]} waves stamped | "
              f"cycles/wave mean (max): loop {m[1]:7.0f} ({mx[1]:7.0f}) of which in s_waitcnt vmcnt {m[2]:7.0f} ({mx[2]:7.0f}) [MFMA {mfma:6.0f}] epilogue {m[3]:6.0f}", flush=True)
