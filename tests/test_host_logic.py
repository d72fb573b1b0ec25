"""CPU: host-side logic -- config parsing, checkpoint / EMA resolution, loader error behaviour, CLI flag
derivation, C-ABI schedule, sharding."""
import argparse
import ctypes
import os

import numpy as np
import pytest
import torch
import yaml

from helpers import get_spec
from open_universe_amd import _lib, inference_utils
from open_universe_amd import config as C
from open_universe_amd import distributed as D
from open_universe_amd import state_dict as S

G = os.path.join(os.path.dirname(__file__), "golden")


def test_config_yaml_roundtrip_and_float_coercion(tmp_path):
    cfg = C.builtin_config("OR16")
    cfg["model"]["diffusion"]["sigma_min"] = "5e-4"  # PyYAML reads 5e-4 as a string
    cfg["model"]["condition_model"]["rate_factors"] = "${model.score_model.rate_factors}"
    p = tmp_path / "config.yaml"
    yaml.safe_dump(cfg, open(p, "w"))
    spec = C.spec_from_config(C.load_config(p))
    assert spec.sigma_min == 5e-4 and spec.cond.rate_factors == [2, 4, 4, 5] and spec.kind == "universe"
    assert spec.diff_kwargs.n_steps == 8 and spec.diff_kwargs.get("epsilon") == 1.3
    with pytest.raises(ValueError):
        C.spec_from_config({"model": dict(cfg["model"], _target_="foo.Bar")})


def test_inference_state_dict_uses_ema_and_ignores_loss_keys():
    spec = get_spec("PP16s")
    sd = S.synthetic_state_dict(spec, seed=0)
    ckpt = S.checkpoint_from_state_dict(spec, sd, with_ema=True, ema_jitter=0.01)
    ckpt["state_dict"]["loss_mpd.foo"] = torch.zeros(3)
    out = S.inference_state_dict(spec, ckpt)
    names = S.parameter_names(spec)
    assert torch.equal(out[names[5]], ckpt["ema"]["shadow_params"][5])
    assert not torch.equal(out[names[5]], sd[names[5]])
    buf = "condition_model.input_mel.mel_spec.mel_scale.fb"
    assert torch.equal(out[buf], sd[buf]) and "loss_mpd.foo" not in out
    bad = dict(ckpt)
    bad["ema"] = dict(ckpt["ema"], shadow_params=ckpt["ema"]["shadow_params"][:-1])
    with pytest.raises(ValueError):
        S.inference_state_dict(spec, bad)
    miss = {"state_dict": {k: v for k, v in sd.items() if "output_conv" not in k}}
    with pytest.raises(KeyError):
        S.inference_state_dict(spec, miss)


def test_load_model_error_behaviour(tmp_path):
    spec = get_spec("PP16s")
    sd = S.synthetic_state_dict(spec, seed=0)
    d = tmp_path / "exp" / "checkpoints"
    d.mkdir(parents=True)
    torch.save(S.checkpoint_from_state_dict(spec, sd), d / "weights.ckpt")
    with pytest.raises(ValueError, match="Could not find the configuration"):  # model_loader.py:45-47
        inference_utils.load_model(d / "weights.ckpt")
    yaml.safe_dump(C.builtin_config("PP16", **{"score_model.n_channels": 8}), open(d / "config.yaml", "w"))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device|HIP"):  # loud failure, no CPU fallback
            inference_utils.load_model(d / "weights.ckpt")
        with pytest.raises(RuntimeError):
            inference_utils.load_model(d / "weights.ckpt", device="cpu")
    with pytest.raises(Exception):  # neither a file nor a reachable HF repo (no network)
        inference_utils.load_model("no-such-org/no-such-model:rev")


def test_add_enhance_arguments_mirrors_signature():
    from open_universe_amd.universe import Universe

    class Dummy:
        enhance = Universe.enhance
        diff_kwargs = get_spec("PP16").diff_kwargs

    parser = inference_utils.add_enhance_arguments(Dummy(), argparse.ArgumentParser())
    args = parser.parse_args(["--n_steps", "16", "--ensemble_stat", "mean"])
    assert args.n_steps == 16 and args.epsilon == 1.3 and args.ensemble_stat == "mean" and args.keep_rms is None
    for k in ("target", "fake_score_snr", "rng", "use_aux_signal", "ensemble", "warm_start"):
        assert hasattr(args, k)
    with pytest.raises(ValueError):
        inference_utils.add_enhance_arguments(object(), argparse.ArgumentParser())


def test_c_abi_schedule_matches_reference_constants(built_lib):
    gold = np.load(os.path.join(G, "schedule.npz"))
    cfg = _lib.make_config(get_spec("PP16"))
    for N in (2, 8, 32, 64):
        sig = (ctypes.c_float * N)()
        eta, beta = ctypes.c_double(), ctypes.c_double()
        assert built_lib.ou_schedule(ctypes.byref(cfg), N, 1.3, sig, ctypes.byref(eta), ctypes.byref(beta)) == 0
        np.testing.assert_allclose(np.array(sig[:]), gold[f"sigma_{N}"], rtol=2e-6)
        assert abs(eta.value - float(gold[f"eta_{N}"])) < 1e-14 and abs(beta.value - float(gold[f"beta_{N}"])) < 1e-14
    assert built_lib.ou_schedule(ctypes.byref(cfg), 1, 1.3, sig, ctypes.byref(eta), ctypes.byref(beta)) == _lib.OU_EINVAL


def test_c_abi_rejects_unsupported_configs(built_lib):
    spec = get_spec("PP16s")
    spec.score.n_channels = 6  # GRU hidden size 48: not a multiple of 64
    spec.cond.n_channels = 6
    n = ctypes.c_size_t()
    cfg = _lib.make_config(spec)
    assert built_lib.ou_packed_bytes(ctypes.byref(cfg), ctypes.byref(n)) == _lib.OU_ENOTIMPL
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.OU_ENOTIMPL)
    cfg = _lib.make_config(get_spec("PP16s"))
    h = ctypes.c_void_p()
    # without a HIP device ou_create must fail loudly (no CPU path)
    if not torch.cuda.is_available():
        dummy = (ctypes.c_float * 4)()
        assert built_lib.ou_packed_bytes(ctypes.byref(cfg), ctypes.byref(n)) == 0
        rc = built_lib.ou_create(ctypes.byref(cfg), dummy, n.value, 0, ctypes.byref(h))
        assert rc == _lib.OU_EHIP


def test_shard_utterances_lpt():
    lengths = [50, 10, 40, 30, 20, 60, 15]
    sh = D.shard_utterances(lengths, 3)
    assert sorted(i for s in sh for i in s) == list(range(7))
    assert sh[0][0] == 5 and sh[1][0] == 0 and sh[2][0] == 2  # longest first, dealt round-robin
    loads = [sum(lengths[i] for i in s) for s in sh]
    assert max(loads) - min(loads) <= max(lengths)


def test_lora_and_weight_norm_removed_checkpoints():
    """state dicts left by LoRA fine-tuning resolve to the same packed blob as the plain merged
    weights; un-mappable EMA lists are refused instead of silently ignored."""
    from helpers import lora_style_state_dict
    from open_universe_amd import _lib

    for name in ("PP16s", "OR16s"):
        spec = get_spec(name)
        sd = S.synthetic_state_dict(spec, seed=2)
        lora_sd, merged = lora_style_state_dict(sd)
        assert any(k.endswith("lora_weight_b") for k in lora_sd) and all(k.startswith("model.") for k in lora_sd)
        got = S.inference_state_dict(spec, {"state_dict": lora_sd})
        assert not any("lora" in k for k in got)
        blob, _ = _lib.pack_weights(spec, got)
        ref_blob, _ = _lib.pack_weights(spec, merged)
        assert torch.equal(blob, ref_blob)
        if spec.score.use_weight_norm:
            # weight-norm folded by the packer (in double) vs folded up front in fp32: same weights to fp32 rounding
            from helpers import split_copy_as_values
            wn_blob, wn_plan = _lib.pack_weights(spec, sd)
            _, plain = lora_style_state_dict(sd, rank=10 ** 6)  # no adapter fits: plain == folded sd
            pb, p_plan = _lib.pack_weights(spec, S.inference_state_dict(spec, plain))
            # (the bf16-split copies hold bit patterns of pieces: compared through the values the pieces add up to)
            pb, wn_blob = split_copy_as_values(pb, p_plan), split_copy_as_values(wn_blob, wn_plan)
            # (the Winograd-domain copies U = G w hold differences of taps: absolute, not relative, agreement there)
            assert torch.allclose(pb, wn_blob, rtol=3e-7, atol=5e-7)
        with pytest.raises(NotImplementedError):
            S.inference_state_dict(spec, {"state_dict": lora_sd, "ema": {"shadow_params": []}})
    # alpha != rank scales the adapter
    a = S.merge_lora({"p.conv.weight": torch.zeros(4, 4, 1), "p.lora_weight_a": torch.ones(4, 2),
                      "p.lora_weight_b": torch.ones(2, 4)}, lora_alpha=1.0)
    assert torch.allclose(a["p.weight"], torch.full((4, 4, 1), 1.0))


def test_checkpoint_reader_executes_nothing(tmp_path):
    """`torch.load(weights_only=False)` on a downloaded file is arbitrary code execution.  The reader takes
    tensors only: a pickle whose `hyper_parameters` would run a callable on load is read without running it."""
    from open_universe_amd.inference_utils import model_loader as ML

    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, (f"touch {marker}",))

    spec = get_spec("PP16s")
    sd = S.synthetic_state_dict(spec, seed=0)
    torch.save({"state_dict": sd, "hyper_parameters": {"x": Evil()}}, tmp_path / "evil.ckpt")
    data = ML.read_checkpoint(tmp_path / "evil.ckpt")
    assert not marker.exists()
    assert set(data["state_dict"]) == set(sd) and all(torch.equal(data["state_dict"][k], sd[k]) for k in sd)
    torch.save({"state_dict": sd, "epoch": 3}, tmp_path / "plain.ckpt")
    assert ML.read_checkpoint(tmp_path / "plain.ckpt")["epoch"] == 3


@pytest.mark.parametrize("spec_str, repo, revision", [("line-corporation/open-universe:plusplus", "line-corporation/open-universe", "plusplus"),
                                                       ("line-corporation/open-universe", "line-corporation/open-universe", None)])
def test_hub_id_is_split_and_both_files_are_fetched(tmp_path, monkeypatch, spec_str, repo, revision):
    """model_loader.py:81-110 of the reference: a path that is not a local file is a Huggingface id `repo[:revision]`;
    `weights.ckpt` and `config.yaml` are fetched with the same repo / revision / token.  The hub itself is unreachable here:
    `hf_hub_download` is replaced by a recorder that serves local files."""
    import huggingface_hub

    from open_universe_amd.inference_utils import model_loader as ML

    served = {"weights.ckpt": tmp_path / "w.ckpt", "config.yaml": tmp_path / "c.yaml"}
    served["weights.ckpt"].write_bytes(b"")
    served["config.yaml"].write_text("model: {}\n")
    calls = []

    def fake(repo_id, filename, revision=None, token=None, **kw):
        calls.append(dict(repo_id=repo_id, filename=filename, revision=revision, token=token))
        return str(served[filename])

    monkeypatch.setattr(huggingface_hub, "hf_hub_download", fake)
    ckpt, cfg = ML._resolve(spec_str, hf_token="tok")
    assert (ckpt, cfg) == (str(served["weights.ckpt"]), str(served["config.yaml"]))
    assert calls == [dict(repo_id=repo, filename="weights.ckpt", revision=revision, token="tok"),
                     dict(repo_id=repo, filename="config.yaml", revision=revision, token="tok")]

    def down(**kw):
        raise OSError("no network")

    monkeypatch.setattr(huggingface_hub, "hf_hub_download", down)
    with pytest.raises(OSError):  # the reference prints a hint and re-raises (model_loader.py:103-107)
        ML._resolve("nobody/nothing:rev", None)


def test_enhance_many_assembles_a_ragged_batch_like_the_serial_loop():
    """Host side of exact batching (Universe.enhance_many without pad_batch), no GPU: the rows, their lengths, the placement of
    every entry's noise -- drawn entry by entry with the shapes of the call on that entry alone, (C_i, 1, L_i + pad_i), from ONE
    shared generator in the serial loop's order -- and the cropping of the outputs."""
    import torch

    from open_universe_amd.universe import Universe

    m = object.__new__(Universe)
    m.tot_ds = 10
    m.device = torch.device("cpu")

    class KW:
        n_steps, epsilon = 3, 1.3

    m.diff_kwargs = KW()
    seen = {}

    def fake_enhance(mix, n_steps, epsilon, target, fake_score_snr, rng, use_aux, keep_rms, ensemble, stat, warm, noise, t_raw=None):
        seen.update(mix=mix.clone(), noise=noise.clone(), t_raw=list(t_raw), n_steps=n_steps)
        return mix * 2.0

    m._enhance = fake_enhance
    m._prep = lambda x: x.to(torch.float32).contiguous()
    sigs = [torch.arange(1, 8, dtype=torch.float32), torch.ones(2, 23), torch.full((10,), 3.0)]  # lengths 7, 23 (2 channels), 10
    g = torch.Generator().manual_seed(5)
    outs = Universe.enhance_many(m, sigs, g)
    assert seen["t_raw"] == [7, 23, 23, 10] and seen["n_steps"] == 3
    assert seen["mix"].shape == (4, 1, 23) and seen["noise"].shape == (3, 4, 1, 30)  # T = 23 + 7
    assert torch.equal(seen["mix"][0, 0, :7], sigs[0]) and not seen["mix"][0, 0, 7:].any()
    # the serial loop's draws from the same generator: entry by entry, step by step, each with its own padded length
    ref = torch.Generator().manual_seed(5)
    exp = torch.zeros(3, 4, 1, 30)
    r = 0
    for s in sigs:
        C, L = (s.shape[0], s.shape[1]) if s.ndim == 2 else (1, s.shape[0])
        Ti = L + (10 - L % 10)
        for k in range(3):
            exp[k, r:r + C, :, :Ti] = torch.randn((C, 1, Ti), generator=ref)
        r += C
    assert torch.equal(seen["noise"], exp)
    assert torch.equal(g.get_state(), ref.get_state())  # the shared generator stands where the serial loop leaves it
    assert not seen["noise"][:, 0, :, 10:].any() and not seen["noise"][:, 3, :, 20:].any()  # zeros behind a row's own pad
    assert [tuple(o.shape) for o in outs] == [(7,), (2, 23), (10,)]
    assert torch.equal(outs[0], 2 * sigs[0]) and torch.equal(outs[1], 2 * sigs[1])
    # equal lengths take the plain path (no t_raw); an empty entry is refused
    seen.clear()

    def fake_plain(mix, n_steps, epsilon, target, fake_score_snr, rng, use_aux, keep_rms, ensemble, stat, warm, noise, t_raw=None):
        seen.update(t_raw=t_raw, B=mix.shape[0])
        return mix

    m._enhance = fake_plain
    Universe.enhance_many(m, [torch.ones(9), torch.zeros(9)], torch.Generator().manual_seed(1))
    assert seen == {"t_raw": None, "B": 2}
    import pytest

    with pytest.raises(ValueError):
        Universe.enhance_many(m, [torch.ones(9), torch.zeros(0)], None)
