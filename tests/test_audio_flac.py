"""FLAC input / output of the CLI without torchaudio: the native decoder (csrc/ou_flac.cpp, `ou_flac_info` / `ou_flac_decode`) against
a bit-level test encoder that can produce every feature of the format (tests/flac_ref_encoder.py), the numpy encoder of the
output side (audio.flac_encode) through the decoder, checksum / signature failures, and the CLI on .flac files."""
import random

import numpy as np
import pytest
import torch

import flac_ref_encoder as E
from open_universe_amd import audio as A
from open_universe_amd.bin import enhance as cli


def _signal(n, bps, seed, ch=1, smooth=True):
    rnd = random.Random(seed)
    full = 1 << (bps - 1)
    out = []
    for c in range(ch):
        x, v, a = [], 0.0, 0.0
        for i in range(n):
            a = 0.95 * a + rnd.gauss(0, 1)
            v = a * full / 40 if smooth else rnd.uniform(-full, full - 1)
            x.append(max(-full, min(full - 1, int(v) + (c * 17))))
        out.append(x)
    return out


def _decode(tmp_path, data, name="t.flac"):
    p = tmp_path / name
    p.write_bytes(data)
    y, fs = A.load(p)
    return y, fs, p


@pytest.mark.parametrize("bps", [8, 12, 16, 20, 24])
def test_every_subframe_type_and_residual_form(tmp_path, bps):
    fs, bs = 16000, 192
    full = 1 << (bps - 1)
    x = _signal(bs * 12, bps, 3 + bps)[0]
    specs = [("verbatim",), ("fixed", 0, 0, 0), ("fixed", 1, 1, 0), ("fixed", 2, 2, 1), ("fixed", 3, 3, 0, (1, 5)),
             ("fixed", 4, 0, 1, (0,)), ("lpc", [3, -1], 3, 1, 0, 0), ("lpc", [1200, -700, 300, -90, 15], 12, 10, 2, 1),
             ("lpc", [14000, -7000] + [50] * 30, 15, 13, 1, 0, (), [3, 9]), ("fixed", 2, 2, 0, (), [0, 14, 7, 1]),
             ("constant",), ("fixed", 1, 0, 0, {"wasted": 3})]
    x[bs * 10:bs * 11] = [x[bs * 10]] * bs                                    # a constant block
    x[bs * 11:] = [(v >> 3) << 3 for v in x[bs * 11:]]                         # three wasted bits
    frames = [E.frame([x[i * bs:(i + 1) * bs]], bps, fs, i, [specs[i]]) for i in range(12)]
    y, fs2, _ = _decode(tmp_path, E.stream([x], bps, fs, frames))
    assert fs2 == fs and y.shape == (1, len(x)) and y.dtype == torch.float32
    assert torch.equal(y[0], torch.tensor(x, dtype=torch.float64).div(full).float())


@pytest.mark.parametrize("stereo", [None, 8, 9, 10])
def test_stereo_decorrelation_block_sizes_and_header_codes(tmp_path, stereo):
    fs, bps = 22050, 16
    sizes = [256, 1152, 100, 4096, 577, 1]                                     # table codes, 8-bit and 16-bit explicit sizes, a last sample
    n = sum(sizes)
    L, R = _signal(n, bps, 11, ch=2)
    R = [max(-32768, min(32767, a + ((i * 7) % 13) - 6)) for i, a in enumerate(L)] if stereo else R
    frames, pos = [], 0
    modes = ["streaminfo", "table", "khz8", "hz16", "tens", "table"]
    for i, bs in enumerate(sizes):
        chans = [L[pos:pos + bs], R[pos:pos + bs]]
        sp = ("fixed", min(2, bs - 1) if bs > 1 else 0, 0, 0) if bs > 2 else ("verbatim",)
        frames.append(E.frame(chans, bps, 22000 if modes[i] == "khz8" else fs if modes[i] != "tens" else 22050, pos, [sp, sp], stereo=stereo,
                              variable=True, rate_mode=modes[i], size_from_streaminfo=(i == 2), force_explicit_block=(i == 0)))
        pos += bs
    data = E.stream([L, R], bps, fs, frames, id3=True, extra_blocks=[(1, b"\x00" * 40), (4, b"\x04\x00\x00\x00test\x00\x00\x00\x00")])
    y, fs2, p = _decode(tmp_path, data)
    assert fs2 == fs and A.channels(p) == 2
    assert torch.equal(y, torch.tensor([L, R], dtype=torch.float64).div(32768).float())


def test_unknown_length_no_signature_and_32_bit_side_channel(tmp_path):
    fs, bps, bs = 48000, 24, 512
    L, R = _signal(bs * 3 + 77, bps, 5, ch=2, smooth=False)                     # full-scale noise: the side channel needs 25 bits
    frames, pos = [], 0
    for i in range(4):
        chans = [L[pos:pos + bs], R[pos:pos + bs]]
        frames.append(E.frame(chans, bps, fs, i, [("verbatim",), ("fixed", 0, 0, 1, (0,))], stereo=10 if i % 2 else 8))
        pos += bs
    y, fs2, _ = _decode(tmp_path, E.stream([L, R], bps, fs, frames, total=0, md5=False))
    assert y.shape == (2, len(L))
    assert torch.equal(y, torch.tensor([L, R], dtype=torch.float64).div(1 << 23).float())


def test_corrupt_streams_are_refused(tmp_path):
    fs, bps, bs = 16000, 16, 576
    x = _signal(bs * 3, bps, 9)[0]
    frames = [E.frame([x[i * bs:(i + 1) * bs]], bps, fs, i, [("fixed", 2, 1, 0)]) for i in range(3)]
    good = E.stream([x], bps, fs, frames)
    _decode(tmp_path, good)
    body = len(good) - len(frames[2]) + 20
    bad = bytearray(good); bad[body] ^= 0x10                                    # a bit inside the last frame's residual
    with pytest.raises(RuntimeError, match="CRC-16|subframe"):
        _decode(tmp_path, bytes(bad), "bad1.flac")
    hdr = len(good) - sum(len(f) for f in frames) + 2
    bad = bytearray(good); bad[hdr] ^= 0x01                                     # a frame-header bit
    with pytest.raises(RuntimeError, match="CRC-8|sync|reserved|block-size"):
        _decode(tmp_path, bytes(bad), "bad2.flac")
    with pytest.raises(RuntimeError, match="ends after|truncated|sync|subframe"):
        _decode(tmp_path, good[:-40], "bad3.flac")
    other = E.stream([[v ^ 1 for v in x]], bps, fs, frames)                     # signature of different audio
    with pytest.raises(RuntimeError, match="MD5"):
        _decode(tmp_path, other, "bad4.flac")
    with pytest.raises(RuntimeError, match="fLaC"):
        _decode(tmp_path, b"RIFF" + good[4:], "bad5.flac")


@pytest.mark.parametrize("bps", [16, 24])
def test_output_encoder_round_trip_and_cli(tmp_path, bps):
    g = torch.Generator().manual_seed(bps)
    t = torch.arange(9000) / 16000.0
    x = torch.stack([0.4 * torch.sin(2 * 3.14159 * 440 * t) + 0.02 * torch.randn(9000, generator=g), torch.zeros(9000),
                     torch.rand(9000, generator=g) * 2 - 1])
    p = tmp_path / "x.flac"
    A.save_flac(p, x, 16000, bps)
    y, fs = A.load(p)
    full = float(1 << (bps - 1))
    want = (torch.clamp(torch.round(x.double() * full), -full, full - 1) / full).float()
    assert fs == 16000 and torch.equal(y, want)
    if bps == 24:
        # the CLI on a directory with a .flac and a .wav file (reference: every suffix of AUDIO_SUFFIXES, output under the input's name)
        from test_cli_cpu import _FakeModel
        src, dst = tmp_path / "in", tmp_path / "out"
        src.mkdir()
        A.save(src / "a.flac", x[:1], 16000)
        A.save(src / "b.wav", x[:1], 16000)
        done = cli.main([str(src), str(dst)], model=_FakeModel())
        assert [q.name for q in done] == ["a.flac", "b.wav"]
        ya, _ = A.load(dst / "a.flac")
        yb, _ = A.load(dst / "b.wav")
        assert float((ya - yb).abs().max()) <= 2.0 ** -23 + 2.0 ** -24          # 24-bit output of half a 24-bit input
