"""GPU (-m gpu): CompressedMagSTFT / CompressedMagSTFTPadded (layers/dyn_range_comp.py:51-225) on
the HIP kernels, against outputs of the reference's own classes (tests/golden/transform.npz) and through the round
trip inverse(forward(x)) == x."""
import os

import numpy as np
import pytest
import torch

import restatement as O
from helpers import TRANSFORM_CASES, get_spec, record, synth_mix

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("case", TRANSFORM_CASES, ids=[c[0] for c in TRANSFORM_CASES])
def test_transform_vs_reference_golden(case):
    from open_universe_amd.layers import CompressedMagSTFT, CompressedMagSTFTPadded

    tag, stft_kw, spec_kw, pad_block = case
    gold = np.load(os.path.join(G, "transform.npz"))
    x = (synth_mix(get_spec("PP16"), 2, 4000)[:, None, :] * 5.0).cuda()
    t = (CompressedMagSTFT(stft_kw, spec_kw) if pad_block is None
         else CompressedMagSTFTPadded(stft_kw, spec_kw, pad_block=pad_block))
    fwd_ref = torch.from_numpy(gold[tag + "_fwd"])
    y = t(x)
    assert y.shape == fwd_ref.shape
    record(f"transform.{tag}.forward", O.si_sdr(fwd_ref, y.cpu()), 90)
    length = None if pad_block else 4000
    inv_ref = torch.from_numpy(gold[tag + "_inv"])
    inv = t.inv(fwd_ref.cuda(), length=length)  # inverse of the REFERENCE's spectrum
    assert inv.shape == inv_ref.shape
    record(f"transform.{tag}.inverse", O.si_sdr(inv_ref, inv.cpu()), 90)
    # round trip through the HIP kernels only (size-independent property)
    back = t.inv(y, length=length)
    n = min(back.shape[-1], x.shape[-1])
    if pad_block is None:
        record(f"transform.{tag}.round_trip", O.si_sdr(x[..., :n].cpu(), back[..., :n].cpu()), 90)
    # inv=True at construction swaps the directions (dyn_range_comp.py:73-74)
    if pad_block is None:
        ti = CompressedMagSTFT(stft_kw, spec_kw, inv=True)
        assert torch.equal(ti(x, inv=True), y)


def test_transform_argument_checks():
    from open_universe_amd.layers import CompressedMagSTFT, CompressedMagSTFTPadded, IdentityTransform

    _, stft_kw, spec_kw, _ = TRANSFORM_CASES[0]
    t = CompressedMagSTFT(stft_kw, spec_kw)
    with pytest.raises(ValueError):
        t(torch.zeros(2, 2, 4000, device="cuda"))          # dyn_range_comp.py:77-78
    with pytest.raises(ValueError):
        t(torch.zeros(2, 100, device="cuda"), inv=True)     # :99-102
    with pytest.raises(ValueError):
        CompressedMagSTFTPadded(stft_kw, spec_kw, pad_block=1000)  # :176-177: not a multiple of the hop
    x = torch.randn(3, 1, 77)
    assert IdentityTransform()(x) is x and IdentityTransform().inv(x) is x
