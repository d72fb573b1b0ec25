"""GPU (-m gpu): EXACT batching of utterances of different lengths (ou_enhance_var, Universe.enhance_many without pad_batch).

Every row of such a batch must be the utterance it would be in a call of its own -- own pad() split (universe.py:219-223), own
normalisation (utils/norm.py:47-87) and mel norm (condition.py:105-106), conv / FIR zero padding right behind its own last
sample on every level, GRU passes over its own frames.  The reference has no such call (its collator pads without a mask,
datasets/datamodule.py:24-42; its CLI loops file by file, bin/enhance.py:173-192), so the yardsticks are
  (a) the file-by-file loop on the HIP path, same per-utterance generators (the reference CLI's semantics), and
  (b) the CPU oracle (pinned against the real reference) run per utterance on identical noise.
Gate: >= 100 dB SI-SDR AND plain SNR for (a) at full width (review target), the usual gates for (b)."""
import ctypes

import pytest
import torch

import restatement as O
from helpers import record, synth_mix, worst
from test_gpu_parity import get_model

pytestmark = pytest.mark.gpu


def _signals(spec, lens, seed=1000):
    return [synth_mix(spec, 1, L, seed=seed + i)[0] for i, L in enumerate(lens)]


def _noise_for(spec, lens, n, seed):
    """Per-utterance CPU noise with the shapes of the call on that utterance alone: list of n tensors (1, 1, T_i)."""
    out = []
    for i, L in enumerate(lens):
        Ti = L + (spec.tot_ds - L % spec.tot_ds)
        g = torch.Generator().manual_seed(seed + i)
        out.append([torch.randn(1, 1, Ti, generator=g) for _ in range(n)])
    return out


def _run_ragged(model, spec, sigs, noise, n_steps, **kw):
    """ou_enhance_var through Universe._enhance with explicit noise -> list of 1-D CPU tensors."""
    lens = [int(s.shape[-1]) for s in sigs]
    B, lm = len(sigs), max(lens)
    T = lm + (spec.tot_ds - lm % spec.tot_ds)
    n_noise = len(noise[0])
    nz = torch.zeros(n_noise, B, 1, T)
    for b in range(B):
        for k in range(n_noise):
            nz[k, b, :, :noise[b][k].shape[-1]] = noise[b][k][0]
    mix = torch.stack([torch.nn.functional.pad(s, (0, lm - s.shape[-1])) for s in sigs])[:, None, :]
    a = dict(use_aux_signal=False, keep_rms=False, warm_start=None)
    a.update(kw)
    out = model._enhance(mix.cuda(), n_steps, None, None, None, None, a["use_aux_signal"], a["keep_rms"], None, "median",
                         a["warm_start"], nz.cuda() if n_noise else None, t_raw=lens).cpu()
    return [out[b, 0, :lens[b]] for b in range(B)], out


def _run_alone(model, sig, noise, n_steps, **kw):
    a = dict(use_aux_signal=False, keep_rms=False, warm_start=None)
    a.update(kw)
    return model._enhance(sig[None, :].cuda(), n_steps, None, None, None, None, a["use_aux_signal"], a["keep_rms"], None,
                          "median", a["warm_start"], [z.cuda() for z in noise]).cpu()[0]


@pytest.mark.parametrize("name", ["PP16s", "PP16m", "OR16s", "PP24s"])
def test_exact_batching_small_models_vs_alone_and_oracle(name):
    """Reduced-width models, lengths from one block to a few dozen incl. T % tot_ds == 0 and T < tot_ds: every row against the
    same utterance alone on the HIP path and against the oracle on identical noise; the padding of every output row is 0."""
    model, spec, sd = get_model(name)
    td = spec.tot_ds
    lens = [td * 23 + 7, td * 9, td * 23 + 7, td * 14 + td // 2, 57, td * 2 - 1, td * 17 + 3]
    N = 3
    sigs = _signals(spec, lens)
    noise = _noise_for(spec, lens, N, 4000)
    rows, full = _run_ragged(model, spec, sigs, noise, N)
    assert full.shape == (len(lens), 1, max(lens))
    for b, L in enumerate(lens):
        assert torch.isfinite(rows[b]).all()
        assert not full[b, 0, L:].any()  # the rest of an output row is zeroed
    alone, orc = [], []
    for b in range(len(lens)):
        alone.append(O.si_sdr(_run_alone(model, sigs[b], noise[b], N), rows[b]))
        ref = O.enhance(sd, spec.to_dict(), sigs[b][None, :], n_steps=N, noise=noise[b])[0]
        orc.append(O.si_sdr(ref, rows[b]))
    record(f"ragged.{name}.worst_row_vs_alone", worst(alone), 80)
    record(f"ragged.{name}.worst_row_vs_oracle", worst(orc))


def test_ragged_invariant_every_activation_is_zero_behind_its_row():
    """What carries the per-row semantics through the network: after a ragged call every named activation is exactly 0 from
    the row's own length on (len = t_pad_b * T_level / T), on every level -- conv outputs, FIR outputs, GRU outputs, x."""
    model, spec, sd = get_model("PP16m")
    td = spec.tot_ds
    lens = [td * 12 + 5, td * 5 + 1, td * 9]
    sigs = _signals(spec, lens)
    _run_ragged(model, spec, sigs, _noise_for(spec, lens, 2, 77), 2)
    T = max(L + (td - L % td) for L in lens)
    tp = [L + (td - L % td) for L in lens]
    names = ["mixn", "x", "cond.in", "cond.mel", "cond.enc0.v", "cond.enc1.h", "cond.enc_sum", "cond.gru0", "cond.gru", "cond.latent",
             "cond.c0", "cond.c2", "cond.c4", "cond.sc1", "cond.aux", "score.in", "score.enc0.v", "score.enc2.v", "score.enc3.h",
             "score.gru", "score.dec0.v", "score.dec2.up", "score.dec2.v", "score.dec4.v"]
    checked = 0
    for nm in names:
        try:
            t = model.tensor(nm)
        except KeyError:
            continue
        Tl = t.shape[-1]
        for b in range(len(lens)):
            lb = tp[b] * Tl // T
            assert lb * T == tp[b] * Tl
            assert not t[b, :, lb:].any(), (nm, b)
            assert t[b, :, :lb].abs().max() > 0, (nm, b)
        checked += 1
    assert checked >= 18, checked


def test_ragged_options_and_multichannel_entries_vs_alone():
    """keep_rms / warm_start / use_aux_signal inside a ragged batch, and enhance_many's public form: (C, L) entries, one
    generator per entry -- against enhance() on each entry alone with the same generator seed."""
    model, spec, sd = get_model("PP16s")
    td = spec.tot_ds
    lens = [td * 15 + 3, td * 6 + 11, td * 11]
    sigs = _signals(spec, lens)
    for tag, kw, N in (("keep_rms", dict(keep_rms=True), 3), ("warm", dict(warm_start=2), 5), ("aux", dict(use_aux_signal=True), 3)):
        n_noise = 0 if kw.get("use_aux_signal") else N - (kw.get("warm_start") or 0)
        noise = _noise_for(spec, lens, n_noise, 900)
        rows, _ = _run_ragged(model, spec, sigs, noise, N, **kw)
        w = worst(O.si_sdr(_run_alone(model, sigs[b], noise[b], N, **kw), rows[b]) for b in range(len(lens)))
        record(f"ragged.PP16s.{tag}.worst_row_vs_alone", w, 80)
    # public form: a 2-channel entry and two mono entries of other lengths, per-entry device generators
    ent = [torch.stack([sigs[0], sigs[0].flip(0)]).cuda(), sigs[1].cuda(), sigs[2].cuda()]
    gens = [torch.Generator(device="cuda").manual_seed(50 + i) for i in range(3)]
    outs = model.enhance_many(ent, gens, n_steps=3)
    assert [tuple(o.shape) for o in outs] == [(2, lens[0]), (lens[1],), (lens[2],)]
    fig = []
    for i, e in enumerate(ent):
        one = model.enhance(e, n_steps=3, rng=torch.Generator(device="cuda").manual_seed(50 + i))
        fig.append(O.si_sdr(one.cpu(), outs[i].cpu()))
    record("ragged.PP16s.enhance_many.worst_entry_vs_alone", worst(fig), 80)
    # a shared generator is consumed exactly as by the serial loop
    g1, g2 = torch.Generator(device="cuda").manual_seed(9), torch.Generator(device="cuda").manual_seed(9)
    model.enhance_many(ent, g1, n_steps=3)
    for e in ent:
        model.enhance(e, n_steps=3, rng=g2)
    assert torch.equal(g1.get_state(), g2.get_state())


def test_ragged_batch_of_equal_lengths_takes_the_plain_path_bit_for_bit():
    model, spec, sd = get_model("PP16s")
    L = spec.tot_ds * 10 + 9
    sigs = _signals(spec, [L, L, L])
    noise = _noise_for(spec, [L, L, L], 3, 21)
    rows, _ = _run_ragged(model, spec, sigs, noise, 3)
    nz = [torch.cat([noise[b][k] for b in range(3)], dim=0) for k in range(3)]
    plain = model._enhance(torch.stack(sigs)[:, None, :].cuda(), 3, None, None, None, None, False, False, None, "median", None,
                           [z.cuda() for z in nz]).cpu()
    assert all(torch.equal(plain[b, 0], rows[b]) for b in range(3))


def test_ou_enhance_var_rejects_bad_lengths():
    model, spec, sd = get_model("PP16s")
    L = spec.tot_ds * 4
    mix = torch.zeros(2, 1, L).cuda()
    T = L + spec.tot_ds
    nz = torch.zeros(2, 2, 1, T).cuda()
    for bad in ([L, 0], [L, L + 1], [L - 1, L - 2]):  # zero length, longer than the buffer, max != T_raw_max
        with pytest.raises(ValueError):
            model._enhance(mix, 2, None, None, None, None, False, False, None, "median", None, nz, t_raw=bad)


@pytest.mark.parametrize("name,secs", [("PP16", [4.0, 1.0, 2.35, 8.0, 3.0, 5.5, 3.99, 6.2]),
                                       ("OR16", [4.0, 1.0, 2.35, 8.0, 3.0, 5.5, 3.99, 6.2]),
                                       ("PP24", [4.0, 1.0, 2.35, 8.0, 3.0, 5.5, 3.99, 6.2])])
def test_exact_batching_full_width_vs_file_by_file_and_oracle(name, secs):
    """Full-width models, 8 utterances of 1 - 8 s (4.0 s and 3.0 s at 16 kHz have T % tot_ds == 0: a whole extra block of
    padding): each row >= 100 dB SI-SDR and plain SNR against the file-by-file loop (HIP) and held against the oracle run per
    utterance on identical noise."""
    model, spec, sd = get_model(name)
    lens = [int(round(s * spec.fs)) for s in secs]
    if name != "PP24":
        assert sum(L % spec.tot_ds == 0 for L in lens) >= 2
    N = 3
    sigs = _signals(spec, lens)
    noise = _noise_for(spec, lens, N, 8800)
    rows, _ = _run_ragged(model, spec, sigs, noise, N)
    alone, orc = [], []
    for b in range(len(lens)):
        alone.append(O.si_sdr(_run_alone(model, sigs[b], noise[b], N), rows[b]))
        ref = O.enhance(sd, spec.to_dict(), sigs[b][None, :], n_steps=N, noise=noise[b])[0]
        orc.append(O.si_sdr(ref, rows[b]))
        print(f"{name} row {b} ({secs[b]} s): vs alone {float(alone[-1]):.1f} dB (snr {alone[-1].snr:.1f}), "
              f"vs oracle {float(orc[-1]):.1f} dB (snr {orc[-1].snr:.1f})")
    wa, wo = worst(alone), worst(orc)
    record(f"ragged.{name}.full.worst_row_vs_file_by_file", wa, 100)
    record(f"ragged.{name}.full.worst_row_vs_oracle", wo, 80)


@pytest.mark.parametrize("name,B,secs", [("PP16", 8, 2.0), ("PP16", 16, 4.0), ("OR16", 3, 1.0), ("PP24", 8, 1.5), ("PP16m", 5, 0.6)])
def test_masks_in_the_epilogues_equal_separate_mask_launches(name, B, secs, steer):
    """The tail masks of a ragged batch live in the epilogues of the conv / FIR / rate-change / in- and out-conv kernels
    (ConvArgs::lens); option mask_fused = 0 launches a separate tail-mask kernel behind every producer instead -- the reference
    form.  Same values everywhere a row is valid, zeros elsewhere: bit-identical outputs, fewer launches.  The shapes cover the
    split-K kernels (batch 3 / 5), the no-split-K family (batch 8) and conv_split_kernel (batch 16 at 4 s)."""
    model, spec, sd = get_model(name)
    g = torch.Generator().manual_seed(3)
    lens = sorted((int(spec.fs * secs * (0.55 + 0.45 * float(torch.rand(1, generator=g)))) for _ in range(B)), reverse=True)
    lens[-1] = (lens[-1] // spec.tot_ds) * spec.tot_ds  # one row with a whole extra block of padding
    sigs = _signals(spec, lens)
    noise = _noise_for(spec, lens, 2, 300)
    steer.set(fuse=0)  # (small batches would run their 64-channel ConvBlock bodies fused with in-kernel masks: next test)
    rows, full = _run_ragged(model, spec, sigs, noise, 2)
    n_fused = sum(model.launch_stats())
    steer.set(mask_fused=0)
    rows0, full0 = _run_ragged(model, spec, sigs, noise, 2)
    n_sep = sum(model.launch_stats())
    steer.unset("mask_fused")
    steer.unset("fuse")
    assert torch.equal(full, full0)
    assert n_fused < n_sep, (n_fused, n_sep)
    # and the invariant holds in the fused form too: a deep tensor of the last score pass is zero behind every row
    t = model.tensor("score.dec1.v")
    T = max(L + (spec.tot_ds - L % spec.tot_ds) for L in lens)
    for b, L in enumerate(lens):
        lb = (L + (spec.tot_ds - L % spec.tot_ds)) * t.shape[-1] // T
        assert not t[b, :, lb:].any() and t[b, :, :lb].abs().max() > 0


@pytest.mark.parametrize("name,B", [("PP16", 2), ("PP16", 3), ("OR16", 2)])
def test_small_ragged_batches_keep_their_fused_convblock_bodies(name, B, steer):
    """Batches of two or three utterances run the 64-channel ConvBlock bodies on conv_chainw_kernel (conv2 + conv3 in one launch,
    the intermediate tile in LDS).  With rows of different lengths the kernel zeroes BOTH stages behind every row's own end
    (ChainArgs::lens) -- the intermediate tile never exists in memory, so no mask launch could do it.  Against the unfused walk
    (option fuse = 0, masks in the conv epilogues): >= 100 dB on every row, fewer launches; against every utterance alone:
    >= 100 dB; the block outputs are zero behind every row."""
    model, spec, sd = get_model(name)
    td = spec.tot_ds
    lens = [int(spec.fs * 2.0) + 77, int(spec.fs * 1.2) // td * td, int(spec.fs * 1.7) + 1][:B]
    sigs = _signals(spec, lens, seed=2200)
    noise = _noise_for(spec, lens, 2, 910)
    rows, full = _run_ragged(model, spec, sigs, noise, 2)
    n_fused = sum(model.launch_stats())
    T = max(L + (td - L % td) for L in lens)
    for nm in ("score.enc1.v", "score.dec3.v"):
        t = model.tensor(nm)
        for b, L in enumerate(lens):
            lb = (L + (td - L % td)) * t.shape[-1] // T
            assert not t[b, :, lb:].any() and t[b, :, :lb].abs().max() > 0, (nm, b)
    steer.set(fuse=0)
    rows0, full0 = _run_ragged(model, spec, sigs, noise, 2)
    n_unfused = sum(model.launch_stats())
    steer.unset("fuse")
    assert n_fused < n_unfused, (n_fused, n_unfused)
    vs_unfused = [O.si_sdr(rows0[b], rows[b]) for b in range(B)]
    vs_alone = []
    for b in range(B):
        vs_alone.append(O.si_sdr(_run_alone(model, sigs[b], noise[b], 2), rows[b]))
        assert not full[b, 0, lens[b]:].any()
    record(f"ragged_fused_bodies.{name}.B{B}.vs_unfused", worst(vs_unfused), 100)
    record(f"ragged_fused_bodies.{name}.B{B}.vs_alone", worst(vs_alone), 100)
