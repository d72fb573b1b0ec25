"""Tuning aid: per-layer time of the k3 / k5 convs at a throughput batch size -- the launcher's own choice ("auto"), the
no-split-K throughput kernel forced (conv_direct3_kernel, cfg 200 + 10 TM + KW), and the kernels of round 2
(OU_CONV_DIRECT=2: split-K direct kernels / LDS kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import json
import torch
from helpers import get_spec
from open_universe_amd import Universe, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

name = sys.argv[1] if len(sys.argv) > 1 else "PP24"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spec = get_spec(name)
model = Universe(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
plan = {c["name"]: c for c in json.loads(model._L.ou_plan_json(model._handle).decode())["convs"]}
T0 = (4 * spec.fs // spec.tot_ds + 1) * spec.tot_ds
pfx = spec.score_prefix
rates = spec.score.rate_factors
layers = []
T = T0
for i in range(len(rates) + 1):
    for cv in ("conv1", "conv2"):
        layers.append((f"{pfx}.encoder.ds_modules.{i}.{cv}", T))
    if i < len(rates):
        T //= rates[i]
# rate-change convs (down: k = s = r; up: transposed conv as r phase GEMMs, KW = 1) and the GRU input projection
T = T0
for i, r in enumerate(rates):
    layers.append((f"{pfx}.encoder.ds_modules.{i}.rate_change_conv", T))
    T //= r
layers.append((f"{pfx}.encoder.gru#l0", T))
nb = len(rates) + 1
for j in range(1, nb):
    layers.append((f"{pfx}.decoder.up_modules.{j}.rate_change_conv", T))
    T *= rates[len(rates) - j]
for lname, Tin in layers:
    L = plan.get(lname)
    if L is None:
        print("missing", lname); continue
    M, Cin, KW = L["M"], L["Cin"], L["KW"]
    stride = L.get("stride", 1)
    Nq = Tin // stride
    flop = 2.0 * M * Cin * KW * Nq * B
    row = []
    tm = 2 if M <= 32 else (3 if (M % 64 and M % 48 == 0) else 4)
    forced = 200 + 10 * tm + KW if (KW in (3, 5) and stride == 1) else 260 + stride
    variants = [("auto", -1, None), ("direct3", forced, None), ("r2", -1, "2")]
    if KW in (3, 5) and stride == 1 and tm == 4:
        variants += [("tm2", 200 + 20 + KW, None), ("tm3", 200 + 30 + KW, None)]
    if KW in (3, 5) and stride == 1:
        variants += [(f"w{t}", 500 + 10 * t + KW, None) for t in ((1, 2, 3) if KW == 3 else (1, 2))] + [("plain", -1, "4")]
    for tag, cfg, env in variants:
        if env is not None:
            os.environ["OU_CONV_DIRECT"] = env
        try:
            ms, used = model.bench_conv(lname, B, Tin, cfg=cfg, with_res=False, iters=10)
            row.append(f"{tag}: {ms*1e3:7.1f} us {flop/ms/1e9:6.1f} TF/s (cfg {used})")
        except Exception as e:
            row.append(f"{tag}: n/a ({str(e)[:40]})")
        os.environ.pop("OU_CONV_DIRECT", None)
    print(f"{lname[-34:]:34s} M={M:4d} Cin={Cin:4d} k{KW} s{stride} up{L.get('up', 1)} T={Tin:6d} B={B} | " + " | ".join(row), flush=True)
