"""GPU (-m gpu): SURVEY 8(f) rank 1 -- a Lightning-format `weights.ckpt` (state_dict + ema.shadow_params,
inference_utils/model_loader.py:117-132, universe.py:130-133, universe_gan.py:136-143) + `config.yaml` through
`load_model`, enhance compared with the oracle run on the EMA weights; and rank 4 -- checkpoints left behind by LoRA
fine-tuning (weight-norm removed, adapters merged or still attached, lora/utils.py:72-89)."""
import pytest
import torch
import yaml

import restatement as O
from helpers import get_spec, lora_style_state_dict, record, synth_mix
from open_universe_amd import config as C
from open_universe_amd import inference_utils
from open_universe_amd import state_dict as S
from test_gpu_parity import noise_list, run_enhance

pytestmark = pytest.mark.gpu


def write_model_dir(tmp_path, name, ckpt):
    base, over = {"PP16s": ("PP16", {"score_model.n_channels": 8}), "OR16s": ("OR16", {"score_model.n_channels": 8})}[name]
    cfg = C.builtin_config(base, **over)
    d = tmp_path / name
    d.mkdir()
    with open(d / "config.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    torch.save(ckpt, d / "weights.ckpt")
    return d / "weights.ckpt"


@pytest.mark.parametrize("name", ["PP16s", "OR16s"])
def test_load_model_lightning_checkpoint_runs_on_ema_weights(name, tmp_path):
    spec = get_spec(name)
    sd = S.synthetic_state_dict(spec, seed=0)
    ckpt = S.checkpoint_from_state_dict(spec, sd, with_ema=True, ema_jitter=0.01, seed=4)
    # what a real checkpoint carries besides the inference tensors: training-only modules and Lightning bookkeeping
    ckpt["state_dict"]["loss_mpd.discriminators.0.convs.0.weight" if name == "PP16s" else "loss_signal.weight"] = torch.zeros(3)
    ckpt["epoch"], ckpt["global_step"] = 3, 1234
    ckpt["hyper_parameters"] = {"fs": spec.fs}
    path = write_model_dir(tmp_path, name, ckpt)
    model, config = inference_utils.load_model(path, device="cuda:0", return_config=True)
    assert model.fs == spec.fs and "model" in config
    ema_sd = S.inference_state_dict(spec, ckpt)
    names = S.parameter_names(spec)
    assert any(not torch.equal(ema_sd[n], sd[n]) for n in names)  # the EMA weights differ from the raw ones
    B, T = 2, spec.tot_ds * 11 + 29
    mix = synth_mix(spec, B, T)
    Tp = T + (spec.tot_ds - T % spec.tot_ds)
    nz = noise_list(19, 4, B, Tp)
    ref = O.enhance(ema_sd, spec.to_dict(), mix, n_steps=4, noise=nz)
    out = run_enhance(model, mix, nz, n_steps=4)
    record(f"loader.{name}.ema_vs_oracle", O.si_sdr(ref, out))
    raw = O.enhance(sd, spec.to_dict(), mix, n_steps=4, noise=nz)
    assert O.si_sdr(raw, out) < 60.0  # ... and the raw weights would not have passed
    # strict=True on a checkpoint without EMA rejects unknown non-training keys (model_loader.py:125-130)
    bad = {"state_dict": dict(sd, **{"mystery.weight": torch.zeros(1)})}
    p2 = tmp_path / "bad"
    p2.mkdir()
    (p2 / "config.yaml").write_text((path.parent / "config.yaml").read_text())
    torch.save(bad, p2 / "weights.ckpt")
    with pytest.raises(RuntimeError):
        inference_utils.load_model(p2 / "weights.ckpt", device="cuda:0", strict=True)
    inference_utils.load_model(p2 / "weights.ckpt", device="cuda:0", strict=False)


def test_load_model_lora_finetuned_checkpoint(tmp_path):
    """A UniverseLoRA-style state dict: `model.` prefix, weight-norm removed (plain `.weight`), Conv1d / Linear adapters
    merged by lora.remove(), ConvTranspose1d adapters still attached -- against the oracle on the merged weights."""
    name = "PP16s"
    spec = get_spec(name)
    sd = S.synthetic_state_dict(spec, seed=2)
    lora_sd, merged = lora_style_state_dict(sd)
    assert any("lora_weight_a" in k for k in lora_sd)
    path = write_model_dir(tmp_path, name, {"state_dict": lora_sd})
    model = inference_utils.load_model(path, device="cuda:0", strict=True)
    B, T = 2, spec.tot_ds * 8 + 3
    mix = synth_mix(spec, B, T)
    Tp = T + (spec.tot_ds - T % spec.tot_ds)
    nz = noise_list(23, 3, B, Tp)
    # the oracle on the merged plain weights (eff_weight takes `.weight` when there is no weight_g / weight_v)
    ref = O.enhance(merged, spec.to_dict(), mix, n_steps=3, noise=nz)
    out = run_enhance(model, mix, nz, n_steps=3)
    record("loader.lora_merged_vs_oracle", O.si_sdr(ref, out))
