"""CPU: the caller side of the path -- audio I/O, the restated torchaudio resampler, file
discovery / argument handling of the `enhance` CLI (reference: open_universe/bin/enhance.py)."""
import math
import wave

import numpy as np
import pytest
import torch

from open_universe_amd import audio as A
from open_universe_amd.bin import enhance as cli


def test_wav_roundtrip_float32_and_pcm16(tmp_path):
    x = torch.randn(2, 1234, generator=torch.Generator().manual_seed(0)) * 0.1
    A.save(tmp_path / "a.wav", x, 22050)
    y, fs = A.load(tmp_path / "a.wav")
    assert fs == 22050 and torch.equal(x, y)
    # a PCM16 file written by the standard library decodes to int / 32768 (torchaudio.load convention)
    pcm = (np.arange(-500, 500) * 60).astype("<i2")
    with wave.open(str(tmp_path / "b.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    y, fs = A.load(tmp_path / "b.wav")
    assert fs == 16000 and y.shape == (1, 1000)
    assert torch.equal(y[0], torch.from_numpy(pcm.astype(np.float32) / 32768.0))
    with pytest.raises(RuntimeError):
        A.load(tmp_path / "c.flac") if A._torchaudio() is None else (_ for _ in ()).throw(RuntimeError())


@pytest.mark.parametrize("fs,target", [(16000, 24000), (24000, 16000), (22050, 16000), (8000, 16000), (16000, 16000)])
def test_resample_properties(fs, target):
    """Length = ceil(new * L / orig) (torchaudio), identity at equal rates, and a tone well below both Nyquist
    frequencies comes out as the same tone at the new rate (away from the edges of the FIR)."""
    L = 4000
    t = torch.arange(L) / fs
    x = torch.sin(2 * math.pi * 440.0 * t)[None]
    y = A.resample(x, fs, target)
    g = math.gcd(fs, target)
    assert y.shape == (1, math.ceil((target // g) * L / (fs // g)))
    if fs == target:
        assert y is x
        return
    tt = torch.arange(y.shape[-1]) / target
    ref = torch.sin(2 * math.pi * 440.0 * tt)[None]
    m = slice(200, y.shape[-1] - 200)
    assert float((y[..., m] - ref[..., m]).abs().max()) < 2e-3


def test_resample_kernel_matches_survey_probe_shapes():
    """SURVEY appendix A.4 (probe of torchaudio's Resample): 1 -> 2 gives a (2, 1, 15) kernel with width 7,
    2 -> 1 a (1, 1, 28) kernel with width 13; and the same numbers as the oracle-side restatement."""
    k12, w12 = A._sinc_kernel(1, 2, "cpu")
    k21, w21 = A._sinc_kernel(2, 1, "cpu")
    assert tuple(k12.shape) == (2, 1, 15) and w12 == 7
    assert tuple(k21.shape) == (1, 1, 28) and w21 == 13
    import restatement as O
    if hasattr(O, "resample_kernel"):
        assert torch.allclose(k12, O.resample_kernel(1, 2)[0], atol=1e-7)


class _FakeModel:
    """enhance(mix) = 0.5 * mix: enough to exercise the script without a GPU."""
    fs = 16000
    device = "cpu"

    class _KW(dict):
        pass

    diff_kwargs = _KW(n_steps=8, epsilon=1.3)

    def __init__(self):
        self.calls = []

    def enhance(self, mix, n_steps: int = None, epsilon: float = None, rng: torch.Generator = None,
                keep_rms: bool = False) -> torch.Tensor:
        self.calls.append((tuple(mix.shape), n_steps, epsilon, rng.initial_seed()))
        return 0.5 * mix


def test_cli_files_arguments_and_seeds(tmp_path):
    src, dst = tmp_path / "in", tmp_path / "out"
    (src / "sub").mkdir(parents=True)
    A.save(src / "b.wav", torch.full((1, 800), 0.25), 16000)
    A.save(src / "sub" / "a.wav", torch.full((2, 1103), 0.5), 22050)
    (src / "notes.txt").write_text("ignored")
    model = _FakeModel()
    done = cli.main([str(src), str(dst), "--seed", "7", "--n_steps", "4"], model=model)
    assert [p.relative_to(dst).as_posix() for p in done] == ["b.wav", "sub/a.wav"]  # sorted, structure retained
    # channels = batch; the second file went through 22050 -> 16000 -> 22050; defaults come from diff_kwargs
    assert model.calls[0] == ((1, 800), 4, 1.3, 7)
    assert model.calls[1][0] == (2, 801) and model.calls[1][3] == 7  # one shared generator (reference semantics)
    y, fs = A.load(dst / "sub" / "a.wav")
    # (like the reference, no trimming after the round trip: ceil(441 * ceil(320 * 1103 / 441) / 320) = 1104)
    assert fs == 22050 and y.shape == (2, 1104)
    assert abs(float(y[:, 300:800].mean()) - 0.25) < 1e-3
    # per-file seeds: file k uses seed + k
    model = _FakeModel()
    cli.main([str(src), str(dst), "--seed", "7", "--per-file-seed"], model=model)
    assert [c[3] for c in model.calls] == [7, 8]
    # single file to an explicit output file
    one = cli.main([str(src / "b.wav"), str(tmp_path / "single.wav")], model=_FakeModel())
    assert one == [tmp_path / "single.wav"] and (tmp_path / "single.wav").exists()


def test_cli_sharding_covers_every_file_once(tmp_path):
    files = []
    for i, n in enumerate([100, 5000, 300, 2500, 40, 900, 1200]):
        p = tmp_path / f"f{i}.wav"
        A.save(p, torch.zeros(1, n), 16000)
        files.append(p)
    for world in (2, 3, 8):
        seen = []
        for rank in range(world):
            seen += [k for k, _ in cli.plan_files(files, world, rank)]
        assert sorted(seen) == list(range(len(files)))
    assert cli.plan_files(files, 1, 0) == list(enumerate(files))
