"""Phase cycle counts of conv_chainw_kernel (OU_CHAIN_TS=<block name>), averaged over blocks and waves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from ctypes import byref, c_size_t, c_void_p
from helpers import get_spec, synth_mix
from open_universe_amd import Universe, state_dict as S, _lib
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option
name = sys.argv[1] if len(sys.argv) > 1 else "score.enc0"
nblk, nw = int(sys.argv[2]) if len(sys.argv) > 2 else 255, 4
os.environ["OU_CHAIN_TS"] = name
spec = get_spec("PP16")
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
def ws_with_slack(B, T):
    if model._ws_key != (B, T):
        n = c_size_t()
        _lib.check(model._L.ou_workspace_bytes(model._handle, B, T, byref(n)), model._handle)
        model._ws = torch.zeros(n.value + (32 << 20), dtype=torch.uint8, device=model.device)
        _lib.check(model._L.ou_workspace_init(model._handle, B, T, c_void_p(model._ws.data_ptr()), c_size_t(model._ws.numel()),
                                              model._stream()), model._handle)
        model._ws_key = (B, T)
        model._cond_key = None
    return model._ws
model._workspace = ws_with_slack
mix = synth_mix(spec, 1, 64000).cuda()
for _ in range(2):
    model.enhance(mix, n_steps=2, rng=torch.Generator(device="cuda").manual_seed(0))
torch.cuda.synchronize()
ws = model._ws
ts = ws[ws.numel() - (16 << 20):].view(torch.int64)[: nblk * nw * 8].view(nblk, nw, 8).cpu().double()
m = ts.mean(dim=(0, 1))
print(f"{name}: cycles/wave: loads issued {m[0]:.0f} | tile + weights landed, staged, barrier {m[1]:.0f} | stage 0: loop {m[2]:.0f} epilogue+barrier {m[3]:.0f} | "
      f"stage 1: {m[4]:.0f} {m[5]:.0f} | stage 2: {m[6]:.0f} {m[7]:.0f} | total {m.sum():.0f}")
