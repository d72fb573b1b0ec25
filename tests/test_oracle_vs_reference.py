"""CPU, build container only: the oracle and the key schema against the live imported reference.
Skipped where /root/reference does not exist (e.g. on the GPU box) -- the committed goldens cover that case."""
import pytest
import torch

import ref_import as R
import restatement as O
from helpers import get_spec, synth_mix
from open_universe_amd import config as C
from open_universe_amd import state_dict as S

pytestmark = pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")


def _ref(name, ref_cfg, width):
    ov = {"score_model.n_channels": width, "condition_model.n_channels": width}
    m, cfg = R.build_reference_model(ref_cfg, ov)
    spec = C.spec_from_config(C.builtin_config(name, **{"score_model.n_channels": width}))
    assert spec.to_dict() == C.spec_from_config({"model": cfg}).to_dict()
    sd = S.synthetic_state_dict(spec, seed=5)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("loss_") for k in missing)
    if m.ema is not None:
        m.ema.shadow_params = [p.clone().detach() for p in m.model_parameters()]
    m.eval()
    return m, spec, sd


@pytest.mark.parametrize("name,ref_cfg", [("PP16", "default"), ("OR16", "universe_original"), ("PP24", "universepp_24k")])
def test_oracle_enhance_matches_live_reference(name, ref_cfg):
    m, spec, sd = _ref(name, ref_cfg, 8)
    mix = synth_mix(spec, 2, spec.tot_ds * 10 + 13)
    with torch.no_grad():
        ref = m.enhance(mix, n_steps=3, rng=torch.Generator().manual_seed(3))
    out = O.enhance(sd, spec.to_dict(), mix, n_steps=3, rng=torch.Generator().manual_seed(3))
    assert O.si_sdr(ref, out) > 100
    # oracle-score diagnostic mode bypasses the network: bit-exact
    tgt = 0.5 * mix[:, None, :]
    with torch.no_grad():
        ref = m.enhance(mix[:, None, :], n_steps=3, target=tgt, fake_score_snr=10.0, rng=torch.Generator().manual_seed(3))
    out = O.enhance(sd, spec.to_dict(), mix[:, None, :], n_steps=3, target=tgt, fake_score_snr=10.0,
                    rng=torch.Generator().manual_seed(3))
    assert torch.equal(ref, out)


def test_buffers_bit_exact_vs_reference():
    m, cfg = R.build_reference_model("default")
    sd = m.state_dict()
    spec = get_spec("PP16")
    for k, shape, is_p in S.model_schema(spec):
        if not is_p:
            assert torch.equal(sd[k], S.buffer_value(k, shape)), k


def test_edm_data_level_and_noisy_ref_match_live_reference():
    """Config options no shipped yaml sets: `edm.data_level_db` (universe.py:176-178) and
    `normalization_kwargs.ref = noisy` (utils/norm.py:83-84, only visible with `target`)."""
    ov = {"score_model.n_channels": 8, "condition_model.n_channels": 8, "edm.data_level_db": -20.0,
          "normalization_kwargs.ref": "noisy"}
    m, cfg = R.build_reference_model("default", ov)
    spec = C.spec_from_config({"model": cfg})
    assert spec.edm_data_level_db == -20.0 and spec.norm_ref == "noisy"
    sd = S.synthetic_state_dict(spec, seed=5)
    m.load_state_dict(sd, strict=False)
    if m.ema is not None:
        m.ema.shadow_params = [p.clone().detach() for p in m.model_parameters()]
    m.eval()
    mix = synth_mix(spec, 2, spec.tot_ds * 6 + 5)
    with torch.no_grad():
        ref = m.enhance(mix, n_steps=3, rng=torch.Generator().manual_seed(3))
    out = O.enhance(sd, spec.to_dict(), mix, n_steps=3, rng=torch.Generator().manual_seed(3))
    assert O.si_sdr(ref, out) > 100
    tgt = 0.5 * mix[:, None, :] + 0.01
    with torch.no_grad():
        ref = m.enhance(mix[:, None, :], n_steps=3, target=tgt, fake_score_snr=10.0, rng=torch.Generator().manual_seed(3))
    out = O.enhance(sd, spec.to_dict(), mix[:, None, :], n_steps=3, target=tgt, fake_score_snr=10.0,
                    rng=torch.Generator().manual_seed(3))
    assert torch.equal(ref, out)


from helpers import TRANSFORM_CASES  # noqa: E402


def test_transform_restatement_matches_live_reference():
    """CompressedMagSTFT(Padded) of the oracle against the reference's classes
    (layers/dyn_range_comp.py), forward and inverse, incl. the Padded variant's double `_pad`."""
    import importlib

    R.install_stubs()
    drc = importlib.import_module("open_universe.layers.dyn_range_comp")
    x = synth_mix(get_spec("PP16"), 2, 4000)[:, None, :] * 5.0
    for tag, stft_kw, spec_kw, pad_block in TRANSFORM_CASES:
        if pad_block is None:
            ref = drc.CompressedMagSTFT(dict(stft_kw), dict(spec_kw))
        else:
            ref = drc.CompressedMagSTFTPadded(dict(stft_kw), dict(spec_kw), pad_block=pad_block)
        kw = dict(n_fft=stft_kw["n_fft"], hop_length=stft_kw["hop_length"], window=ref.stft_window,
                  pad_block=pad_block, **spec_kw)
        y_ref = ref(x)
        y = O.compressed_mag_stft(x, **kw)
        assert y.shape == y_ref.shape and torch.equal(y, y_ref), tag
        length = None if pad_block else 4000
        assert torch.equal(O.compressed_mag_stft(y, inv=True, length=length, **kw), ref.inv(y_ref, length=length)), tag
