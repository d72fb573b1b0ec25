"""Tuning aid: where a conv_direct4_kernel launch spends its time (per-wave phase stamps, OU_TS): block start spread, per-wave
prologue / main loop / epilogue cycles against the MFMA cycles of the wave.   python tools/d4_ts.py [B]"""
import os, sys
os.environ["OU_TS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from ctypes import byref, c_float, c_int32, c_size_t, c_void_p
from helpers import get_spec
from open_universe_amd import Universe, state_dict as S, _lib
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
spec = get_spec("PP16")
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
p = "_edm_model"
T0 = 64160
layers = [(p + ".encoder.gru#l0", T0 // 160, 128, 1), (p + ".decoder.up_modules.2.rate_change_conv", T0 // 32, 64, 1),
          (p + ".decoder.up_modules.3.rate_change_conv", T0 // 8, 32, 1), (p + ".encoder.ds_modules.2.rate_change_conv", T0 // 8, 32, 4),
          (p + ".decoder.signal_cond_proj.0", T0 // 160, 128, 1),
          (p + ".encoder.ds_modules.4.conv1", T0 // 160, 128, 0), (p + ".encoder.ds_modules.4.conv2", T0 // 160, 128, 0)]
ws = torch.zeros(1 << 29, dtype=torch.uint8, device="cuda")
tail = ws[ws.numel() - (16 << 20):].view(torch.int64)
for lname, Tin, slots, R in layers:
    for cfg in ((-1, 323, 343, 322, 342) if R else (-1,)):
        tm, wk = ((cfg - 300) // 10, 1 << ((cfg - 300) % 10)) if cfg > 0 else (0, 0)
        ms, used = c_float(), c_int32()
        tail.zero_()
        try:
            _lib.check(model._L.ou_bench_conv(model._handle, lname.encode(), B, Tin, cfg, -1, 0, 3, c_void_p(ws.data_ptr()),
                                              c_size_t(ws.numel()), model._stream(), byref(ms), byref(used)), model._handle)
        except Exception as e:
            continue
        torch.cuda.synchronize()
        u = used.value
        if 600 <= u < 800:  # conv_direct4w_kernel: 600 / 700 + 10 TM + KW
            wk, tm, kw = (8 if u < 700 else 4), (u % 100) // 10, u % 10
            ts = tail[: 8192 * 8 * 8].view(-1, 8).cpu().double()
            ts = ts[ts[:, 7] > 0]
            t0, t1 = ts[:, 0], ts[:, 7]
            m, mx = ts.mean(dim=0), ts.max(dim=0).values
            mfma = slots / wk * 2 * (kw + 1) * tm * 32
            print(f"{lname[-40:]:40s} cfg{u} {ms.value * 1e3:6.1f} us/launch | {ts.shape[0]} waves, span {(t1.max() - t0.min()).item() / 100:5.1f} us "
                  f"| cycles/wave mean (max): prologue {m[1]:5.0f} ({mx[1]:5.0f}) loop {m[2]:6.0f} ({mx[2]:6.0f}) [MFMA {mfma:6.0f}] "
                  f"(of which in s_waitcnt vmcnt {m[6]:6.0f}) A^T + slab write {m[3]:5.0f} ({mx[3]:5.0f}) barrier wait {m[4]:5.0f} ({mx[4]:5.0f}) reduce + store {m[5]:5.0f} ({mx[5]:5.0f})", flush=True)
            continue
        if not 300 <= u < 400:
            print(f"{lname[-40:]:40s} cfg{u} {ms.value * 1e3:6.1f} us/launch (not conv_direct4_kernel)")
            continue
        tm, wk = (u - 300) // 10, 1 << ((u - 300) % 10)
        nw = 4 if wk == 1 else wk
        ts = tail[: 8192 * 8 * 8].view(-1, 8).cpu().double()
        ts = ts[ts[:, 7] > 0]
        if ts.shape[0] == 0:
            print(f"{lname[-40:]:40s} cfg{u}: no stamps")
            continue
        t0, t1 = ts[:, 0], ts[:, 7]
        span = (t1.max() - t0.min()).item() * 10
        starts = (t0 - t0.min()) * 10
        m = ts.mean(dim=0)
        mx = ts.max(dim=0).values
        mfma = slots / wk * 4 * tm * R * 32
        q = lambda x, f: torch.quantile(x, f).item()
        print(f"{lname[-40:]:40s} cfg{u} {ms.value * 1e3:6.1f} us/launch | {ts.shape[0]} waves, span {span / 1e3:5.1f} us, wave start p50 "
              f"{q(starts, .5) / 1e3:4.1f} p90 {q(starts, .9) / 1e3:4.1f} max {starts.max().item() / 1e3:4.1f} us | cycles/wave mean (max): "
              f"prologue {m[1]:5.0f} ({mx[1]:5.0f}) loop {m[2]:6.0f} ({mx[2]:6.0f}) [MFMA {mfma:6.0f}] epilogue {m[3]:5.0f} ({mx[3]:5.0f})", flush=True)
