"""
Generates the golden fixtures in this directory by importing the REAL reference (/root/reference, build
container only -- it never travels to the GPU box) with framework stubs (oracle/ref_import.py), loading the
seeded synthetic weights (recipe: open_universe_amd.state_dict.synthetic_state_dict) and running the
reference's own `Universe.enhance` / networks on CPU.

Fixtures are DATA only: inputs are regenerated from seeds, outputs are stored as float32 arrays.
    keys_<cfg>.json      ordered (key, shape, is_parameter) of the reference state dict + model_parameters() order
    small_<cfg>.npz      reduced-width models: conditioner / score-net / enhance outputs (several option sets)
    full_PP16.npz        UNIVERSE++ 16 kHz full size, 4 s, 8 steps (the headline configuration)
    schedule.npz         sampler constants for N in {2, 8, 32, 64}
    stress_<cfg>.npz     second weight draw (seed 1) and the "stress" draw (seed 3, gain 1.3: activations and GRU
                         pre-activations several times larger) of the reduced-width models
    full_PP16_n64.npz    BASELINE configs[2] per-utterance shape: UNIVERSE++ 16 kHz, 4 s, 64 steps
    full_OR16_n32.npz    BASELINE configs[3]: original UNIVERSE 16 kHz, 4 s, 32 steps, 2 utterances
    full_PP24_varlen.npz BASELINE configs[4]: UNIVERSE++ 24 kHz, 8 steps, variable-length batch of 8 right-zero-padded
                         to its longest member (datasets/datamodule.py:24-42); rows 0 / 7 (longest / shortest) stored
    transform.npz        CompressedMagSTFT(Padded) forward / inverse of the reference's own classes (4 parameter sets)
    loud_<cfg>.npz       inputs loud enough that the peak guard (universe.py:356-357) divides: row 0 at the usual level, row 1
                         60 x louder; with keep_rms (the RMS restore puts row 1 far above full scale, the guard fires on that row
                         only) and without (normalize_batch removes the level; the guard's state is asserted, not assumed)
    fullstress_<cfg>.npz FULL-WIDTH models (PP16, OR16) on three hard weight draws -- s3g13 (seed 3, gain 1.3), s5g15 / s5g14 (seed 5,
                         gain 1.5 / 1.4: the harshest that stays finite in the reference), s7g12t4 (seed 7, gain 1.2, heavy-tailed
                         weight-norm gains) --, 4 s, 8 steps, two utterances (the HIP tests
                         run row 0 alone = the batch-1 dispatch, and rows 0-1 inside a batch of 16 = the throughput dispatch).
                         Beside every output the REFERENCE'S OWN SELF-AGREEMENT on those weights as scalars: 1 thread vs 8
                         threads and fp32 vs fp64 (SI-SDR / SNR in dB, row 0) -- the yardstick for the HIP path's figures
Run:  python tests/golden/make_golden.py [base] [stress] [configs] [transform] [loud] [fullstress]      (default: base)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import ref_import as R  # noqa: E402
from helpers import SMALL, get_spec, synth_mix  # noqa: E402
from open_universe_amd import state_dict as S  # noqa: E402

REF_CFG = {"PP16": "default", "OR16": "universe_original", "PP24": "universepp_24k"}


STRESS = {"s1g1": dict(seed=1, gain=1.0), "s3g13": dict(seed=3, gain=1.3)}


def varlen_lengths(fs=24000, n=8, seed=5):
    """L_i = fs * U(1, 8) s, manual_seed(5), sorted descending."""
    u = torch.rand(n, generator=torch.Generator().manual_seed(seed))
    return sorted((int(fs * (1.0 + 7.0 * float(v))) for v in u), reverse=True)


def build(name, seed=0, gain=1.0, tail=0.0):
    base, over = SMALL.get(name, (name, {}))
    ov = {}
    for k, v in over.items():
        ov[k] = v
        ov[k.replace("score_model", "condition_model")] = v
    m, cfg = R.build_reference_model(REF_CFG[base], ov)
    spec = get_spec(name)
    sd = S.synthetic_state_dict(spec, seed=seed, gain=gain, tail=tail)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("loss_") for k in missing)
    if m.ema is not None:
        m.ema.shadow_params = [p.clone().detach() for p in m.model_parameters()]
    m.eval()
    return m, spec, sd


def normalized_input(spec, mix, Tp):
    T = mix.shape[-1]
    xin = torch.nn.functional.pad(mix[:, None, :], ((Tp - T) // 2, (Tp - T) - (Tp - T) // 2))
    xin = (xin - xin.mean(dim=(1, 2), keepdim=True))
    return xin * (10 ** (spec.level_db / 20) / xin.std(dim=(1, 2), keepdim=True).clamp(min=1e-5))


def make_stress():
    for name in ("PP16s", "PP16m", "OR16s", "PP24s"):
        out = {}
        for tag, kw in STRESS.items():
            m, spec, sd = build(name, **kw)
            B, T = 2, spec.tot_ds * 12 + 5
            mix = synth_mix(spec, B, T)
            Tp = T + (spec.tot_ds - T % spec.tot_ds)
            with torch.no_grad():
                xin = normalized_input(spec, mix, Tp)
                cond, aux, lat = m.condition_model(xin, x_wav=xin, train=True)
                sig = torch.tensor([0.3, 1.7])
                xs = noise_list(11, 1, B, Tp)[0] * sig[:, None, None]
                out[tag + "_score"] = m.score_model(xs, sig, cond).numpy()
                out[tag + "_latent"] = lat.numpy()
                out[tag + "_cond_last"] = cond[-1].numpy()
            out[tag + "_enh"] = enhance_with_noise(m, mix, noise_list(7, 4, B, Tp), n_steps=4).numpy()
            out["B"], out["T"] = B, T
        np.savez_compressed(os.path.join(HERE, f"stress_{name}.npz"), **out)
        print("stress", name, {k: getattr(v, "shape", v) for k, v in out.items()})


def make_configs():
    # C3 (per-utterance shape): UNIVERSE++ 16 kHz, 64 steps
    m, spec, sd = build("PP16")
    T = 64000
    Tp = T + (spec.tot_ds - T % spec.tot_ds)
    enh = enhance_with_noise(m, synth_mix(spec, 1, T), noise_list(303, 64, 1, Tp), n_steps=64)
    np.savez_compressed(os.path.join(HERE, "full_PP16_n64.npz"), enh=enh.numpy().astype(np.float32), T=T)
    print("C3", enh.shape, float(enh.std()))
    # C4: original UNIVERSE, 32 steps
    m, spec, sd = build("OR16")
    enh = enhance_with_noise(m, synth_mix(spec, 2, T), noise_list(404, 32, 2, Tp), n_steps=32)
    np.savez_compressed(os.path.join(HERE, "full_OR16_n32.npz"), enh=enh.numpy().astype(np.float32), T=T)
    print("C4", enh.shape, float(enh.std()))
    # C5: UNIVERSE++ 24 kHz, variable-length batch of 8 (right zero padding, no mask)
    m, spec, sd = build("PP24")
    lens = varlen_lengths(spec.fs)
    Tm = lens[0]
    batch = torch.stack([torch.nn.functional.pad(synth_mix(spec, 1, L, seed=1000 + i)[0], (0, Tm - L))
                         for i, L in enumerate(lens)])
    Tp = Tm + (spec.tot_ds - Tm % spec.tot_ds)
    enh = enhance_with_noise(m, batch, noise_list(505, 8, len(lens), Tp), n_steps=8)
    np.savez_compressed(os.path.join(HERE, "full_PP24_varlen.npz"), lens=np.array(lens),
                        row0=enh[0, :lens[0]].numpy().astype(np.float32),
                        row7=enh[7, :lens[7]].numpy().astype(np.float32))
    print("C5", lens, enh.shape, float(enh.std()))


# (gain 1.6 overflows in the reference itself at full width; OR16 -- no weight-norm, no EDM pre-conditioning -- already at 1.5)
FULL_STRESS = {"PP16": {"s3g13": dict(seed=3, gain=1.3), "s5g15": dict(seed=5, gain=1.5), "s7g12t4": dict(seed=7, gain=1.2, tail=0.4)},
               "OR16": {"s3g13": dict(seed=3, gain=1.3), "s5g14": dict(seed=5, gain=1.4), "s7g12t4": dict(seed=7, gain=1.2, tail=0.4)}}


def _db(ref, est):
    """(SI-SDR, plain SNR) in dB of est against ref, float64."""
    r, e = ref.double().flatten(), est.double().flatten()
    a = float((r * e).sum() / (r * r).sum())
    si = 10 * np.log10(float(((a * r) ** 2).sum() / ((e - a * r) ** 2).sum().clamp(min=1e-300)))
    sn = 10 * np.log10(float((r ** 2).sum() / ((e - r) ** 2).sum().clamp(min=1e-300)))
    return si, sn


def make_fullstress():
    """Full-width stress goldens + the reference's self-agreement ('what's weak' 1(i)-(ii))."""
    T, N, B = 64000, 8, 2
    for name in ("PP16", "OR16"):
        out = {"T": T, "B": B, "n_steps": N}
        for tag, kw in FULL_STRESS[name].items():
            m, spec, sd = build(name, **kw)
            Tp = T + (spec.tot_ds - T % spec.tot_ds)
            mix = synth_mix(spec, B, T)
            nz = noise_list(2100 + kw["seed"], N, B, Tp)
            torch.set_num_threads(8)
            enh = enhance_with_noise(m, mix, nz, n_steps=N)
            assert torch.isfinite(enh).all()
            out[tag + "_enh"] = enh.numpy().astype(np.float32)
            # self-agreement of the reference on row 0: 8 threads vs 1 thread, batch of 2 vs alone, fp32 vs fp64
            nz0 = [z[:1] for z in nz]
            one8 = enhance_with_noise(m, mix[:1], nz0, n_steps=N)
            torch.set_num_threads(1)
            one1 = enhance_with_noise(m, mix[:1], nz0, n_steps=N)
            torch.set_num_threads(8)
            m64 = m.double()
            one64 = enhance_with_noise(m64, mix[:1].double(), [z.double() for z in nz0], n_steps=N)
            m.float()
            out[tag + "_self_threads"] = np.array(_db(one8, one1))
            out[tag + "_self_batch"] = np.array(_db(one8, enh[:1]))
            out[tag + "_self_fp64"] = np.array(_db(one64, one8))
            print("fullstress", name, tag, "std", float(enh.std()), "peak", float(enh.abs().max()),
                  "threads", out[tag + "_self_threads"], "batch", out[tag + "_self_batch"], "fp64", out[tag + "_self_fp64"],
                  flush=True)
        np.savez_compressed(os.path.join(HERE, f"fullstress_{name}.npz"), **out)


LOUD_GAINS = (1.0, 60.0)


def make_loud():
    """Last link of enhance (universe.py:349-357): `x * (mix_rms / x_rms)` then `x / max|x|` where the peak exceeds 1."""
    for name in ("PP16s", "OR16s", "PP24s"):
        m, spec, sd = build(name)
        B, T = 2, spec.tot_ds * 12 + 5
        mix = synth_mix(spec, B, T) * torch.tensor(LOUD_GAINS)[:, None]
        Tp = T + (spec.tot_ds - T % spec.tot_ds)
        out = {"B": B, "T": T}
        for tag, kw in {"keep_rms": dict(n_steps=3, keep_rms=True), "plain": dict(n_steps=3)}.items():
            enh = enhance_with_noise(m, mix, noise_list(9, 3, B, Tp), **kw)
            out["enh_" + tag] = enh.numpy()
        pk = np.abs(out["enh_keep_rms"]).max(axis=-1)
        assert pk[0] < 0.9 and abs(pk[1] - 1.0) < 1e-6, pk  # the guard divided row 1 and left row 0 alone
        np.savez_compressed(os.path.join(HERE, f"loud_{name}.npz"), **out)
        print("loud", name, "peaks keep_rms", pk, "plain", np.abs(out["enh_plain"]).max(axis=-1))


def noise_list(seed, n, B, T):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, 1, T, generator=g) for _ in range(n)]


class NoiseGen:
    """Stands in for torch.Generator in the reference's randn(): we monkey-patch torch.randn instead."""


def enhance_with_noise(m, mix, noise, **kw):
    """Run the reference's enhance with pre-drawn noise by temporarily replacing torch.randn."""
    it = iter(noise)
    real = torch.randn

    def fake(*a, **k):
        return next(it).clone()

    torch.randn = fake
    try:
        with torch.no_grad():
            return m.enhance(mix, **kw)
    finally:
        torch.randn = real


def make_transform():
    """CompressedMagSTFT / CompressedMagSTFTPadded of the reference (layers/dyn_range_comp.py) on a seeded signal."""
    import importlib

    from helpers import TRANSFORM_CASES

    R.install_stubs()
    drc = importlib.import_module("open_universe.layers.dyn_range_comp")
    x = synth_mix(get_spec("PP16"), 2, 4000)[:, None, :] * 5.0
    out = {}
    for tag, stft_kw, spec_kw, pad_block in TRANSFORM_CASES:
        t = (drc.CompressedMagSTFT(dict(stft_kw), dict(spec_kw)) if pad_block is None
             else drc.CompressedMagSTFTPadded(dict(stft_kw), dict(spec_kw), pad_block=pad_block))
        y = t(x)
        out[tag + "_fwd"] = y.numpy()
        out[tag + "_inv"] = t.inv(y, length=None if pad_block else 4000).numpy()
    np.savez_compressed(os.path.join(HERE, "transform.npz"), **out)
    print("transform", {k: v.shape for k, v in out.items()})


def main():
    torch.set_num_threads(8)
    what = set(sys.argv[1:]) or {"base"}
    if "transform" in what:
        make_transform()
    if "stress" in what:
        make_stress()
    if "configs" in what:
        make_configs()
    if "loud" in what:
        make_loud()
    if "fullstress" in what:
        make_fullstress()
    if "base" not in what:
        return
    # ---- key schema
    for name, ref in REF_CFG.items():
        m, cfg = R.build_reference_model(ref)
        pn = set(n for n, _ in m.named_parameters())
        keys = [[k, list(v.shape), k in pn] for k, v in m.state_dict().items() if not k.startswith("loss_")]
        ids = {id(p): n for n, p in m.named_parameters()}
        order = [ids[id(p)] for p in m.model_parameters()]
        json.dump({"keys": keys, "parameter_order": order}, open(os.path.join(HERE, f"keys_{name}.json"), "w"))
        print("keys", name, len(keys))
    # ---- small models
    for name in ("PP16s", "PP16m", "OR16s", "PP24s"):
        m, spec, sd = build(name)
        B, T = 2, spec.tot_ds * 20 + 37
        mix = synth_mix(spec, B, T)
        Tp = T + (spec.tot_ds - T % spec.tot_ds)
        out = {"B": B, "T": T}
        with torch.no_grad():
            xin = torch.nn.functional.pad(mix[:, None, :], ((Tp - T) // 2, (Tp - T) - (Tp - T) // 2))
            xin = (xin - xin.mean(dim=(1, 2), keepdim=True))
            xin = xin * (10 ** (spec.level_db / 20) / xin.std(dim=(1, 2), keepdim=True).clamp(min=1e-5))
            cond, aux, lat = m.condition_model(xin, x_wav=xin, train=True)
            for j, c in enumerate(cond):
                out[f"cond{j}"] = c.numpy()
            out["aux"] = aux.numpy()
            out["latent"] = lat.numpy()
            sig = torch.tensor([0.3, 1.7])
            xs = noise_list(11, 1, B, Tp)[0] * sig[:, None, None]
            out["score"] = m.score_model(xs, sig, cond).numpy()
        opts = {"plain": dict(n_steps=4), "keep_rms": dict(n_steps=3, keep_rms=True),
                "ens_median": dict(n_steps=3, ensemble=3, ensemble_stat="median"),
                "ens_mean": dict(n_steps=3, ensemble=2, ensemble_stat="mean"),
                "ens_sigmed": dict(n_steps=3, ensemble=3, ensemble_stat="signal_median")}
        if spec.use_signal_decoupling:
            opts["warm"] = dict(n_steps=5, warm_start=2)
            opts["aux"] = dict(n_steps=4, use_aux_signal=True)
        for tag, kw in opts.items():
            E = kw.get("ensemble") or 1
            nz = noise_list(7, kw["n_steps"], B * E, Tp)
            out["enh_" + tag] = enhance_with_noise(m, mix, nz, **kw).numpy()
        np.savez_compressed(os.path.join(HERE, f"small_{name}.npz"), **out)
        print("small", name, {k: getattr(v, "shape", v) for k, v in out.items()})
    # ---- full-size headline config
    m, spec, sd = build("PP16")
    T = 64000
    mix = synth_mix(spec, 1, T)
    Tp = T + (spec.tot_ds - T % spec.tot_ds)
    nz = noise_list(1028282, 8, 1, Tp)
    enh = enhance_with_noise(m, mix, nz, n_steps=8)
    np.savez_compressed(os.path.join(HERE, "full_PP16.npz"), enh=enh.numpy().astype(np.float32), T=T)
    print("full PP16", enh.shape, float(enh.std()))
    # ---- sampler constants (universe.py:301-311)
    sch = {}
    for N in (2, 8, 32, 64):
        delta_t = 1.0 / (N - 1)
        gamma = (spec.sigma_max / spec.sigma_min) ** -delta_t
        eps = 1.3
        sch[f"eta_{N}"] = 1 - gamma ** eps
        sch[f"beta_{N}"] = (1 - gamma ** (2 * (eps - 1.0))) ** 0.5
        time = torch.linspace(0, 1, N).flip(dims=[0])
        sch[f"sigma_{N}"] = m.get_std_dev(time).numpy()
    np.savez(os.path.join(HERE, "schedule.npz"), **sch)


if __name__ == "__main__":
    main()
