"""Tuning aid: where a conv_direct_kernel launch spends its time.  Per-wave phase stamps (OU_TS) of the deep-level PP16
layers at batch 1: ramp (spread of block start times), per-wave prologue / first-data wait / main loop / drain / epilogue
cycles, and the launch's span against its MFMA-bound minimum."""
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

name = sys.argv[1] if len(sys.argv) > 1 else "PP16"
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
p = "_edm_model"
T0 = 64160
layers = [(p + ".encoder.ds_modules.2.conv1", T0 // 8), (p + ".encoder.ds_modules.2.conv2", T0 // 8),
          (p + ".encoder.ds_modules.3.conv1", T0 // 32), (p + ".encoder.ds_modules.3.conv2", T0 // 32),
          (p + ".encoder.ds_modules.4.conv1", T0 // 160), (p + ".encoder.ds_modules.4.conv2", T0 // 160),
          (p + ".encoder.gru#l0", T0 // 160), (p + ".decoder.up_modules.1.rate_change_conv", T0 // 160)]
ws = torch.zeros(1 << 29, dtype=torch.uint8, device="cuda")
tail = ws[ws.numel() - (16 << 20):].view(torch.int64)
for lname, Tin in layers:
    for cfg in (-1,):
        ms, used = c_float(), c_int32()
        tail.zero_()
        _lib.check(model._L.ou_bench_conv(model._handle, lname.encode(), 1, Tin, cfg, -1, 1, 3, c_void_p(ws.data_ptr()),
                                          c_size_t(ws.numel()), model._stream(), byref(ms), byref(used)), model._handle)
        torch.cuda.synchronize()
        ts = tail[: 4096 * 8 * 8].view(4096, 8, 8).cpu().double()
        on = ts[:, 0, 7] > 0
        ts = ts[on]
        nwv = int((ts[0, :, 7] > 0).sum().item()) if ts.shape[0] else 8   # 8 waves per block, or 4 (the four-slice variants)
        ts = ts[:, :nwv]
        nb = ts.shape[0]
        if nb == 0:
            print(f"{lname[-40:]:40s} cfg{used.value}: no stamps (not a direct kernel)")
            continue
        t0, t1 = ts[..., 0], ts[..., 7]
        span = (t1.max() - t0.min()).item() * 10      # ns
        starts = (t0[:, 0] - t0.min()) * 10
        dur = (t1.max(dim=1).values - t0.min(dim=1).values) * 10
        m = ts.mean(dim=(0, 1))
        q = lambda x, f: torch.quantile(x, f).item()
        print(f"{lname[-40:]:40s} cfg{used.value} {ms.value*1e3:6.1f} us/launch | {nb} blocks, span {span/1e3:5.1f} us, block start p50 "
              f"{q(starts, .5)/1e3:4.1f} p90 {q(starts, .9)/1e3:4.1f} max {starts.max().item()/1e3:4.1f} us, block duration p50 "
              f"{q(dur, .5)/1e3:4.1f} p90 {q(dur, .9)/1e3:4.1f} us | cycles/wave: issue {m[1]:5.0f} first-data {m[2]:5.0f} "
              f"loop {m[3]:6.0f} drain {m[4]:5.0f} epilogue {m[5]:5.0f}", flush=True)
