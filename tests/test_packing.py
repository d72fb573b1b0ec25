"""CPU: the C++ packer (weight-norm fold, binomial-FIR fold, transposed-conv phase split, GRU layouts) is
checked by unpacking the blob and comparing a torch emulation of the generic conv against the oracle."""
import json

import pytest
import torch
import torch.nn.functional as F

import restatement as O
from helpers import emulate_conv, get_spec, plan_convs, unpack_conv
from open_universe_amd import _lib
from open_universe_amd import state_dict as S


@pytest.mark.parametrize("name", ["PP16s", "OR16s", "PP24s"])
def test_pack_all_convs(built_lib, name):
    spec = get_spec(name)
    sd = S.synthetic_state_dict(spec, seed=3)
    blob, plan = _lib.pack_weights(spec, sd)
    assert blob.numel() * 4 == _lib.packed_bytes(spec)
    convs = plan_convs(plan)
    g = torch.Generator().manual_seed(0)
    n_checked = 0
    for nm, L in convs.items():
        kind = L["kind"]
        if kind == 4:  # GRU projection
            pfx, lay = nm.split("#l")
            H, I = L["Cout"] // 6, L["Cin"]
            x = torch.randn(2, I, 9, generator=g)
            y = emulate_conv(blob, L, x)
            for d, sfx in enumerate(("", "_reverse")):
                k = f"_l{lay}{sfx}"
                ref = F.conv1d(x, sd[pfx + ".weight_ih" + k][:, :, None], sd[pfx + ".bias_ih" + k])
                bhh = sd[pfx + ".bias_hh" + k].clone()
                bhh[2 * H:] = 0
                ref = ref + bhh.view(1, -1, 1)
                assert torch.allclose(y[:, d * 3 * H:(d + 1) * 3 * H], ref, atol=2e-5, rtol=1e-5), nm
        elif kind == 3:  # st conv lowered to space-to-depth + 1x1
            R = L["rate"]
            C = L["Cin"] // R
            x = torch.randn(2, C, 3 * R, generator=g)
            ref = O.prelu_conv(sd, nm, x, stride=R)
            xs = F.prelu(x, sd[nm + ".prelu.weight"]).view(2, C, 3, R).permute(0, 1, 3, 2).reshape(2, C * R, 3)
            y = emulate_conv(blob, L, xs, act=False)
            assert torch.allclose(y, ref, atol=2e-5, rtol=1e-4), nm
        elif kind == 1:  # down
            r = L["rate"]
            x = torch.randn(2, L["Cin"], 6 * r, generator=g)
            ref = O.prelu_conv(sd, nm, x, stride=r)
            y = emulate_conv(blob, L, x)
            assert torch.allclose(y, ref, atol=2e-5, rtol=1e-4), (nm, float((y - ref).abs().max()))
        elif kind == 2:  # up
            r = L["rate"]
            x = torch.randn(2, L["Cin"], 7, generator=g)
            ref = O.prelu_conv(sd, nm, x, stride=r, transpose=True)
            y = emulate_conv(blob, L, x)
            assert torch.allclose(y, ref, atol=2e-5, rtol=1e-4), (nm, float((y - ref).abs().max()))
        else:  # plain 'same' conv
            x = torch.randn(2, L["Cin"], 11, generator=g)
            if L["act"]:
                ref = O.prelu_conv(sd, nm, x, same=True)
            else:
                ref = F.conv1d(x, O.eff_weight(sd, nm), sd[nm + ".bias"], padding="same")
            y = emulate_conv(blob, L, x)
            assert torch.allclose(y, ref, atol=2e-5, rtol=1e-4), nm
        n_checked += 1
    assert n_checked == len(convs) and n_checked > 60


def test_taps_innermost_weight_copy_equals_the_tap_major_one(built_lib):
    """Stride-1 k3 / k5 layers with Cin % 16 == 0 carry a second copy [Cin][Mp][4 | 8] for conv_direct2_kernel (Cin % 64)
    and conv_direct3_kernel (Cin % 16): same numbers as the [Cin/CK][KW][CK][Mp] layout, zero in the padding taps and rows."""
    spec = get_spec("PP16m")
    sd = S.synthetic_state_dict(spec, seed=5)
    blob, plan = _lib.pack_weights(spec, sd)
    n = 0
    for nm, L in plan_convs(plan).items():
        if not L["KWP"]:
            assert not (L["stride"] == 1 and L["up"] == 1 and L["KW"] in (3, 5) and L["Cin"] % 16 == 0), nm
            continue
        assert L["KW"] in (3, 5) and L["KWP"] == (4 if L["KW"] == 3 else 8) and L["Cin"] % 16 == 0
        Cin, KW, KWP, Mp, M = L["Cin"], L["KW"], L["KWP"], L["Mp"], L["M"]
        W, _, _ = unpack_conv(blob, L)                                     # [M][Cin][KW]
        wd = blob[L["wd_off"]: L["wd_off"] + Cin * Mp * KWP].view(Cin, Mp, KWP)
        assert torch.equal(wd[:, :M, :KW].permute(1, 0, 2), W), nm
        assert not wd[:, :, KW:].any() and not wd[:, M:, :].any(), nm
        n += 1
    assert n >= 20


def test_winograd_domain_weight_copy_reproduces_the_convolution(built_lib):
    """The third copy U = G w (F(2, 3): 4 values, F(2, 5): 6 values per (row, channel), ou_model.cpp) together with the
    kernel's B^T / A^T must BE the convolution: y[2p + j] = sum_k w[k] d[2p + j + k].  Checked in double on the packed fp32
    values: an error in any of the three matrices -- they live in two source files -- shows here, without a GPU."""
    BT = {3: [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
          5: [[2, -3, -4, 3, 2, 0], [0, -2, 1, 5, 2, 0], [0, -2, 5, -1, -2, 0], [0, 2, 1, -2, -1, 0], [0, 1, -2, -1, 2, 0],
              [0, 2, -3, -4, 3, 2]]}
    AT = {3: [[1, 1, 1, 0], [0, 1, -1, -1]], 5: [[1, 1, 1, 1, 1, 0], [0, 1, -1, 0.5, -2, 1]]}
    spec = get_spec("PP16m")
    sd = S.synthetic_state_dict(spec, seed=5)
    blob, plan = _lib.pack_weights(spec, sd)
    g = torch.Generator().manual_seed(3)
    n = 0
    for nm, L in plan_convs(plan).items():
        if not L["KWP"]:
            continue
        Cin, KW, KWP, Mp, M = L["Cin"], L["KW"], L["KWP"], L["Mp"], L["M"]
        W, _, _ = unpack_conv(blob, L)                                     # [M][Cin][KW]
        wu = blob[L["wu_off"]: L["wu_off"] + Cin * Mp * KWP].view(Cin, Mp, KWP)
        assert not wu[:, :, KW + 1:].any() and not wu[:, M:, :].any(), nm
        U = wu[:, :M, :KW + 1].permute(1, 0, 2).double()                   # [M][Cin][KW + 1]
        d = torch.randn(Cin, KW + 1, generator=g).double()                 # one tile: KW + 1 input samples per channel
        V = d @ torch.tensor(BT[KW], dtype=torch.float64).T                # [Cin][KW + 1]
        Mx = (U * V[None]).sum(dim=1)                                      # [M][KW + 1]: reduction over the channels
        y = Mx @ torch.tensor(AT[KW], dtype=torch.float64).T               # [M][2]
        ref = torch.stack([(W.double() * d[None, :, j:j + KW]).sum(dim=(1, 2)) for j in range(2)], dim=1)
        assert torch.allclose(y, ref, rtol=0, atol=2e-5 * float(ref.abs().max())), (nm, float((y - ref).abs().max()))
        n += 1
    assert n >= 20


def test_bf16_split_weight_copy_is_the_fp32_weight_exactly(built_lib):
    """The fourth copy (conv_split_kernel, ou_split_pack.h): every fp32 weight as three bf16 pieces laid out as MFMA A fragments
    [Cin/16][KW][Mp/32][3][64 lanes][8]: lane l holds row 32 mt + (l & 31), channels 16 cc + 8 (l >> 5) + j.  hi + mid + lo must BE
    the fp32 weight (24 significand bits = 3 x 8: the split is exact), hi must be the bf16 rounding of it, padding rows zero --
    without a GPU."""
    spec = get_spec("PP16")
    sd = S.synthetic_state_dict(spec, seed=5)
    blob, plan = _lib.pack_weights(spec, sd)
    n = 0
    for nm, L in plan_convs(plan).items():
        if not L.get("ws_on"):
            assert not (L["KWP"] and L["M"] % 64 == 0), nm
            continue
        Cin, KW, Mp, M = L["Cin"], L["KW"], L["Mp"], L["M"]
        W, _, _ = unpack_conv(blob, L)                                     # [M][Cin][KW]
        nfl = Cin * KW * Mp * 3 // 2
        raw = blob[L["ws_off"]: L["ws_off"] + nfl].view(torch.int16)       # bf16 bit patterns
        fr = raw.view(Cin // 16, KW, Mp // 32, 3, 64, 8)
        pieces = (fr.to(torch.int32) << 16).view(torch.float32)            # bf16 -> fp32, exact
        # [cc][k][mt][piece][half][row32][j] -> [piece][row][ci][k]
        pc = pieces.view(Cin // 16, KW, Mp // 32, 3, 2, 32, 8).permute(3, 2, 5, 0, 4, 6, 1).reshape(3, Mp, Cin, KW)
        total = pc[0].double() + pc[1].double() + pc[2].double()
        assert torch.equal(total[:M].float(), W) and torch.equal(total[:M], W.double()), nm
        assert not pc[:, M:].any(), nm
        assert torch.equal(pc[0][:M], W.to(torch.bfloat16).float()), nm    # hi = round-to-nearest-even bf16 of the weight
        assert float((pc[1][:M].abs() - W.abs() * 2.0 ** -8).max()) <= 0, nm
        n += 1
    assert n >= 20


def test_pack_errors(built_lib):
    spec = get_spec("PP16s")
    sd = S.synthetic_state_dict(spec, seed=0)
    bad = dict(sd)
    del bad["_edm_model.encoder.ds_modules.0.conv1.conv.weight_v"]
    with pytest.raises(KeyError):
        _lib.pack_weights(spec, bad)
    bad = dict(sd)
    bad["condition_model.input_conv.bias"] = torch.zeros(3)
    with pytest.raises(ValueError):
        _lib.pack_weights(spec, bad)


def test_library_exports_every_declared_symbol(built_lib):
    import re, os
    inc = os.path.join(os.path.dirname(__file__), "..", "include")
    declared = set(re.findall(r"\b(ou_[a-z_]+)\s*\(", open(os.path.join(inc, "ouniverse.h")).read()))
    assert declared == set(_lib.EXPORTED_SYMBOLS)  # the drop-in boundary
    tuning = set(re.findall(r"\b(ou_[a-z_]+)\s*\(", open(os.path.join(inc, "ouniverse_tuning.h")).read()))
    assert tuning == set(_lib.TUNING_SYMBOLS) and not (tuning & declared)  # measurement entry points kept apart
    for s in declared | tuning:
        assert hasattr(built_lib, s), s


def test_default_build_carries_no_experiment_kernels(built_lib):
    """`make` (what __graft_entry__.build() runs) leaves the kernels no default dispatch rule can select out of the library:
    conv_block3_kernel (one launch per deep ConvBlock: measured slower end to end), round 1's gru_cluster_kernel, round 3's
    conv_chain_kernel (superseded by conv_chainw_kernel), round 6's conv_splitw_kernel (measured slower) and the
    4-wave / 64x64 split-K configs of conv_mfma_kernel are in `make EXPERIMENTS=1` builds only.  The device code objects
    inside the .so name their kernels in clear text."""
    import os

    so = os.path.join(os.path.dirname(__file__), "..", "open-universe_amd", "lib", "libouniverse.so")
    blob = open(so, "rb").read()
    assert b"conv_direct2_kernel" in blob and b"gru_ring_kernel" in blob and b"conv_direct4_kernel" in blob
    if b"+experiments" in built_lib.ou_version():
        pytest.skip("an EXPERIMENTS build")
    for name in (b"conv_block3_kernel", b"gru_cluster_kernel", b"conv_chain_kernel", b"conv_splitw_kernel"):
        assert name not in blob, name
    # conv_mfma_kernel<TM, TN, WM, WN, WK, ...>: no instantiation with a 4-way split of the reduction (WK = 4)
    assert b"conv_mfma_kernelILi1ELi2ELi1ELi1ELi4E" not in blob and b"conv_mfma_kernelILi1ELi1ELi1ELi1ELi4E" not in blob


def test_library_reads_no_environment_variable(built_lib):
    """Since ABI 5 every tuning / test switch is a typed option of the handle (ou_set_option, include/ouniverse.h); the shared
    library does not import getenv at all, so nothing outside the caller's code can change which kernels run or what they
    compute.  The option table is self-describing: unique keys, a one-line description and a default each."""
    import os
    import subprocess

    so = os.path.join(os.path.dirname(__file__), "..", "open-universe_amd", "lib", "libouniverse.so")
    und = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und and "environ" not in und
    names = _lib.option_names()
    dflt = _lib.option_defaults()
    assert len(names) == built_lib.ou_option_count() >= 25 and set(names) == set(dflt)
    assert all(len(doc) > 10 for doc in names.values())
    assert dflt["conv_direct"] == 5 and dflt["split"] == -1 and dflt["wino"] == 1 and dflt["gru_dbg"] == 0 and dflt["dbg"] == 0
    assert built_lib.ou_option_name(-1) is None and built_lib.ou_option_name(len(names)) is None
    # the Python side reads exactly two variables of its own, and the rendezvous / launcher variables of torch.distributed
    pkg = os.path.join(os.path.dirname(__file__), "..", "open-universe_amd")
    seen = set()
    import re
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                seen |= set(re.findall(r"[\"'](OU_[A-Z_]+)[\"']", open(os.path.join(root, f)).read()))
    assert seen <= {"OU_LIBRARY", "OU_UNSAFE_PICKLE", "OU_", "OU_CHAIN_TS"}, seen


def test_blob_without_the_split_copy_is_a_quarter_smaller(built_lib):
    """ou_config.no_split_copy: handles that never run a batch of 8 or more utterances per call leave the bf16-split copy out of
    the blob (what the RCCL broadcast moves and every rank keeps resident); every other slot keeps its value."""
    spec = get_spec("PP16m")
    sd = S.synthetic_state_dict(spec, seed=2)
    full, plan_f = _lib.pack_weights(spec, sd)
    lean, plan_l = _lib.pack_weights(spec, sd, split_copy=False)
    assert full.numel() * 4 == _lib.packed_bytes(spec) and lean.numel() * 4 == _lib.packed_bytes(spec, split_copy=False)
    assert 0.70 < lean.numel() / full.numel() < 0.80
    cf, cl = plan_convs(plan_f), plan_convs(plan_l)
    assert any(c["ws_on"] for c in cf.values()) and not any(c["ws_on"] for c in cl.values())
    for nm, c in cf.items():
        n = c["Cin"] * c["KW"] * c["Mp"]
        assert torch.equal(full[c["w_off"]: c["w_off"] + n], lean[cl[nm]["w_off"]: cl[nm]["w_off"] + n]), nm
