"""GPU bring-up helper: runs conditioner / score net / enhance through the C ABI on cuda:0 and prints the
SI-SDR of every named intermediate against the CPU oracle.  (Test infrastructure; not a product path.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import restatement as O
from helpers import get_spec, synth_mix
from open_universe_amd import Universe, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option


def cmp(name, ref, got):
    got = got.detach().cpu()
    if ref.shape != got.shape:
        print(f"  {name:24s} SHAPE MISMATCH ref {tuple(ref.shape)} got {tuple(got.shape)}")
        return -999
    s = O.si_sdr(ref, got)
    flag = "" if s > 80 else "   <<<<<<"
    print(f"  {name:24s} {s:7.1f} dB   rms ref {float(ref.std()):.4g} got {float(got.std()):.4g}{flag}")
    return s


def run(name, B=2, T=None, n_steps=4, seed=0):
    spec = get_spec(name)
    sd = S.synthetic_state_dict(spec, seed=seed)
    sdict = spec.to_dict()
    if T is None:
        T = spec.tot_ds * 20
    print(f"=== {name}: B={B} T={T}")
    model = Universe(spec, state_dict=sd, device="cuda:0")
    mix = synth_mix(spec, B, T)
    xin = O.normalize(mix[:, None, :], spec.level_db)
    # ---- conditioner
    taps = {}
    c_ref, y_ref, h_ref = O.conditioner_network(sd, "condition_model", sdict, xin, taps=taps)
    cond, aux, lat = model.condition_model(xin.cuda(), train=True)
    cmp("cond.mel", taps["mel"], model.tensor("cond.mel"))
    cmp("cond.x_mel", taps["x_mel"], model.tensor("cond.melblock.v"))
    for i in range(len(spec.score.rate_factors) - 1):
        cmp(f"cond.st{i}", taps[f"st{i}"], model.tensor(f"cond.st{i}"))
    cmp("cond.enc_sum", taps["enc_sum"], model.tensor("cond.enc_sum"))
    cmp("cond.gru", taps["gru"], model.tensor("cond.gru"))
    cmp("cond.latent", h_ref, lat)
    for j, (a, b) in enumerate(zip(c_ref, cond)):
        cmp(f"cond.c{j}", a, b)
    cmp("cond.aux", y_ref, aux)
    if spec.use_signal_decoupling:
        cmp("aux_to_wav", O.aux_to_wav(sd, sdict, y_ref), model.aux_to_wav())
    # ---- score network
    g = torch.Generator().manual_seed(5)
    sig = torch.tensor([0.3, 1.7, 0.01, 4.0][:B])
    xs = torch.randn(xin.shape, generator=g) * sig[:, None, None]
    taps = {}
    if spec.edm_noise is not None:
        w = O.edm_weights(sdict, sig)
        O.score_network(sd, "_edm_model", sdict, w["in"][:, None, None] * xs, w["noise"] * sig, c_ref, taps=taps)
    else:
        O.score_network(sd, "score_model", sdict, xs, sig, c_ref, taps=taps)
    s_ref = O.score_model(sd, sdict, xs, sig, c_ref)
    s_hip = model.score_model(xs.cuda(), sig)
    cmp("score.in", taps["input_conv"], model.tensor("score.in"))
    nb = len(spec.score.rate_factors) + int(spec.score.extra_conv_block)
    for i in range(nb):
        cmp(f"score.enc{i}.c1", taps[f"enc{i}.v"] * 0 + 1, model.tensor(f"score.enc{i}.c1") * 0 + 1) if False else None
        cmp(f"score.enc{i}.v", taps[f"enc{i}.v"], model.tensor(f"score.enc{i}.v"))
        if i < len(spec.score.rate_factors):
            cmp(f"score.enc{i}.h", taps[f"enc{i}.h"], model.tensor(f"score.enc{i}.h"))
    if spec.score.extra_conv_block:
        ref = (taps["gru"] + taps[f"enc{nb-1}.v"]) / 2 ** 0.5
    else:
        ref = taps["gru"]
    cmp("score.gru(+res)", ref, model.tensor("score.gru"))
    for j in range(nb):
        cmp(f"score.dec{j}.v", taps[f"dec{j}.v"], model.tensor(f"score.dec{j}.v"))
    cmp("score (final)", s_ref, s_hip)
    # ---- enhance
    Tp = T + (spec.tot_ds - T % spec.tot_ds)
    noise = [torch.randn(B, 1, Tp, generator=g) for _ in range(n_steps)]
    t0 = time.time()
    e_ref = O.enhance(sd, sdict, mix, n_steps=n_steps, noise=noise)
    t1 = time.time()
    e_hip = model._enhance(mix.cuda(), n_steps, None, None, None, None, False, False, None, "median", None,
                           [z.cuda() for z in noise])
    torch.cuda.synchronize()
    t2 = time.time()
    cmp("enhance", e_ref, e_hip)
    print(f"  oracle {t1-t0:.2f}s hip(first call) {t2-t1:.3f}s launches {model.launch_stats()}")




def timing(name="PP16", B=1, T=64000, n_steps=8, iters=5, check=0):
    spec = get_spec(name)
    sd = S.synthetic_state_dict(spec, seed=0)
    model = Universe(spec, state_dict=sd, device="cuda:0")
    mix = synth_mix(spec, B, T).cuda()
    rng = torch.Generator(device="cuda").manual_seed(1028282)
    model.check_status = bool(check)
    for _ in range(2):
        model.enhance(mix, n_steps=n_steps, rng=rng)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.time()
        model.enhance(mix, n_steps=n_steps, rng=rng)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"TIMING {name} B={B} T={T} N={n_steps}: median {med*1e3:.2f} ms  min {ts[0]*1e3:.2f} ms  "
          f"RTF {B*T/spec.fs/med:.1f}x  utt/s {B/med:.2f}  launches {model.launch_stats()}")
    if os.environ.get("OU_STALL_DIAG"):
        d = model._ws[:128].view(torch.int32).cpu().tolist()
        print("   all iterations ms:", [round(t * 1e3, 1) for t in ts], "| ring-GRU net activations", d[20], "first:",
              d[21:30], "non-plain workgroups", d[30], "system-scope mode", d[31])


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "timing":
        timing(*(args[1:2] or ["PP16"]), **{k: int(v) for k, v in (a.split("=") for a in args[2:])})
    elif args and args[0] == "full":
        run(args[1], B=int(args[2]), T=int(args[3]), n_steps=int(args[4]))
    else:
        for n in args or ["PP16s", "PP16m", "OR16s", "PP24s"]:
            run(n)
