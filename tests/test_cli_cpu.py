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
        A.load(tmp_path / "c.mp3") if A._torchaudio() is None else (_ for _ in ()).throw(RuntimeError())  # (.flac: test_audio_flac.py)


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
    # the checkpoint-buffer restatement (state_dict.sinc_resample_kernel) is a separately written function
    from open_universe_amd import state_dict as S
    assert torch.allclose(k12, S.sinc_resample_kernel(1, 2), atol=1e-7)
    assert torch.allclose(k21, S.sinc_resample_kernel(2, 1), atol=1e-7)


def test_torchaudio_restatements_pinned_by_definition():
    """torchaudio is absent from the image, so its two pieces of arithmetic on the path (HTK mel filterbank, condition.py
    :75-81; sinc_interp_hann resampling kernels, alias_free_act.py:21-22) are restated from its documentation.  Pinned
    here against an independent float64 numpy evaluation of the published formulas and hand-computed values."""
    import math

    import numpy as np

    from open_universe_amd import state_dict as S

    # ---- HTK filterbank, melscale_fbanks(n_freqs, 0, sr/2, n_mels, sr, norm=None, mel_scale="htk")
    for n_fft, n_mels in ((640, 80), (960, 128)):
        n_freqs, sr = n_fft // 2 + 1, 24000
        fb = S.mel_filterbank(n_freqs, n_mels, sr).double().numpy()
        mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
        imel = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
        pts = imel(np.linspace(mel(0.0), mel(sr / 2), n_mels + 2))
        freqs = np.linspace(0, sr // 2, n_freqs)
        ref = np.zeros((n_freqs, n_mels))
        for m in range(n_mels):
            lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
            up = (freqs - lo) / (ce - lo)
            down = (hi - freqs) / (hi - ce)
            ref[:, m] = np.maximum(0.0, np.minimum(up, down))
        assert np.abs(fb - ref).max() < 2e-5
        assert fb.shape == (n_freqs, n_mels) and fb.min() >= 0.0 and fb.max() <= 1.0
        assert np.all(fb[0] == 0.0) and np.abs(fb[-1]).max() < 1e-5  # DC and Nyquist sit on the outer edges
    # hand-computed: mel(12 kHz) = 2595 log10(1 + 12000/700) = 3266.34...; first centre of the 80-band bank
    assert abs(2595.0 * math.log10(1.0 + 12000.0 / 700.0) - 3266.3412) < 1e-3
    c1 = 700.0 * (10.0 ** ((3266.3412 / 81.0) / 2595.0) - 1.0)  # = 25.5 Hz < one 37.5 Hz bin: band 0 has <= 1 bin
    assert 25.0 < c1 < 26.0 and (S.mel_filterbank(321, 80)[:, 0] > 0).sum() <= 2

    # ---- sinc_interp_hann kernels (lowpass_filter_width 6, rolloff 0.99)
    for orig, new, width in ((1, 2, 7), (2, 1, 13)):
        k = S.sinc_resample_kernel(orig, new).double().numpy()
        base = min(orig, new) * 0.99
        idx = np.arange(-width, width + orig, dtype=np.float64) / orig
        ref = []
        for ph in range(new):
            t = np.clip((-ph / new + idx) * base, -6.0, 6.0)
            ref.append(np.sinc(t) * np.cos(t * np.pi / 12.0) ** 2 * base / orig)  # np.sinc(x) = sin(pi x) / (pi x)
        assert np.abs(k[:, 0, :] - np.stack(ref)).max() < 1e-7
    k12 = S.sinc_resample_kernel(1, 2)
    assert abs(float(k12[0, 0, 7]) - 0.99) < 1e-6        # phase 0, t = 0: sinc(0) * hann(0) * base/orig
    assert abs(float(S.sinc_resample_kernel(2, 1)[0, 0, 13]) - 0.495) < 1e-6
    assert abs(float(k12[0].sum()) - 1.0) < 0.02 and abs(float(k12[1].sum()) - 1.0) < 0.02  # unit DC gain per phase


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


def test_cli_groups_consecutive_files_of_equal_rate_any_length():
    """--batch-size groups consecutive files of one sample rate whatever their lengths (exact batching keeps every row's own
    geometry; --pad-batch only changes what the call does with them)."""
    from open_universe_amd.bin import enhance as cli

    todo = [(k, f"f{k}") for k in range(6)]
    infos = {0: (16000, 5), 1: (16000, 5), 2: (16000, 5), 3: (16000, 7), 4: (8000, 7), 5: (8000, 7)}
    g = cli.group_files(todo, infos, 2, False)
    assert [[k for k, _ in grp] for grp in g] == [[0, 1], [2, 3], [4, 5]]
    g = cli.group_files(todo, infos, 4, True)
    assert [[k for k, _ in grp] for grp in g] == [[0, 1, 2, 3], [4, 5]]


class _NoisyFake(_FakeModel):
    """enhance(x) = x + noise from the generator it is handed, one draw of the input's shape: a wrong generator state, a
    wrong grouping or a wrong order changes the output."""
    tot_ds = 1

    def __init__(self):
        super().__init__()
        self.batches = []

    def enhance(self, mix, n_steps: int = None, epsilon: float = None, rng: torch.Generator = None,
                keep_rms: bool = False) -> torch.Tensor:
        return mix + torch.randn(mix.shape, generator=rng)

    def enhance_many(self, sigs, rngs, pad_batch=False, **kw):
        self.batches.append([int(s.shape[-1]) for s in sigs])
        if not isinstance(rngs, (list, tuple)):
            rngs = [rngs] * len(sigs)
        return [s + torch.randn(s.shape, generator=g) for s, g in zip(sigs, rngs)]

    def advance_generator_like_enhance(self, rng, channels, length, **kw):
        torch.randn((channels, length), generator=rng)


def test_cli_length_sorted_window_keeps_the_serial_noise(tmp_path):
    """--batch-size with the read-ahead window: files are grouped by length inside the window (a call costs what its longest
    row costs) -- and every file still gets the noise the file-by-file loop would give it from the ONE shared generator: its
    generator starts from the state taken in front of it in processing order.  Outputs bit-equal to the serial run."""
    src = tmp_path / "in"
    src.mkdir()
    lens = [900, 200, 800, 300, 700, 100, 600, 250, 150]
    for i, n in enumerate(lens):
        A.save(src / f"f{i}.wav", 0.001 * torch.randn(2 if i == 2 else 1, n, generator=torch.Generator().manual_seed(i)), 16000)
    serial = _NoisyFake()
    cli.main([str(src), str(tmp_path / "o1"), "--seed", "11"], model=serial)
    batched = _NoisyFake()
    cli.main([str(src), str(tmp_path / "o2"), "--seed", "11", "--batch-size", "2", "--batch-window", "6"], model=batched)
    for i in range(len(lens)):
        a, _ = A.load(tmp_path / "o1" / f"f{i}.wav")
        b, _ = A.load(tmp_path / "o2" / f"f{i}.wav")
        assert torch.equal(a, b), i
    # window 1: files 0..5 sorted by length -> (900, 800), (700, 300), (200, 100); window 2: (600, 250), (150)
    assert batched.batches == [[900, 800], [700, 300], [200, 100], [600, 250], [150]]
    # per-file seeds: no state juggling needed, same grouping
    pf = _NoisyFake()
    cli.main([str(src), str(tmp_path / "o3"), "--seed", "11", "--batch-size", "2", "--batch-window", "6", "--per-file-seed"], model=pf)
    assert pf.batches == batched.batches
