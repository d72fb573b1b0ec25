"""Audio file I/O and resampling for the `enhance` CLI (the caller side of the hot path).

The reference CLI uses `torchaudio.load / save / functional.resample` (bin/enhance.py:77-80,180-192).  torchaudio
is used when it is installed; otherwise .wav files are read / written with the standard library + numpy (the only
format that needs no codec), and `resample` is restated from torchaudio's documented algorithm
(`sinc_interp_hann`, lowpass_filter_width 6, rolloff 0.99 -- the defaults the reference relies on).
"""
import math
import wave
from pathlib import Path

import numpy as np
import torch

AUDIO_SUFFIXES = [".wav", ".mp3", ".flac"]  # bin/enhance.py:33


def _torchaudio():
    try:
        import torchaudio  # noqa: F401

        return torchaudio
    except Exception:
        return None


def can_decode(path):
    """True when `load` can read this file here: anything with torchaudio, .wav without."""
    return _torchaudio() is not None or Path(path).suffix.lower() == ".wav"


def load(path):
    """-> (float32 tensor (channels, T) in [-1, 1], sample rate): torchaudio.load semantics."""
    ta = _torchaudio()
    if ta is not None:
        return ta.load(str(path))
    path = Path(path)
    if path.suffix.lower() != ".wav":
        raise RuntimeError(f"{path}: only .wav can be decoded without torchaudio")
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise RuntimeError(f"{path}: not a RIFF/WAVE file")
        fmt = None
        data = None
        while True:
            ck = f.read(8)
            if len(ck) < 8:
                break
            cid, size = ck[:4], int.from_bytes(ck[4:], "little")
            body = f.read(size + (size & 1))[:size]
            if cid == b"fmt ":
                fmt = body
            elif cid == b"data":
                data = body
        if fmt is None or data is None:
            raise RuntimeError(f"{path}: missing fmt / data chunk")
    tag = int.from_bytes(fmt[0:2], "little")
    ch = int.from_bytes(fmt[2:4], "little")
    fs = int.from_bytes(fmt[4:8], "little")
    bits = int.from_bytes(fmt[14:16], "little")
    if tag == 0xFFFE and len(fmt) >= 26:  # WAVE_FORMAT_EXTENSIBLE: sub-format GUID starts with the real tag
        tag = int.from_bytes(fmt[24:26], "little")
    if tag == 3 and bits == 32:
        x = np.frombuffer(data, dtype="<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(data, dtype="<f8").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(data, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / 8388608.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(data, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise RuntimeError(f"{path}: unsupported WAV encoding (format tag {tag}, {bits} bits)")
    n = (x.size // ch) * ch
    return torch.from_numpy(x[:n].reshape(-1, ch).T.copy()), fs


def channels(path):
    """Number of channels of an audio file from its header alone (no sample is decoded)."""
    ta = _torchaudio()
    if ta is not None:
        return int(ta.info(str(path)).num_channels)
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise RuntimeError(f"{path}: not a RIFF/WAVE file")
        while True:
            ck = f.read(8)
            if len(ck) < 8:
                raise RuntimeError(f"{path}: missing fmt chunk")
            cid, size = ck[:4], int.from_bytes(ck[4:], "little")
            if cid == b"fmt ":
                return int.from_bytes(f.read(4)[2:4], "little")
            f.seek(size + (size & 1), 1)


def save(path, audio, fs):
    """audio: float tensor (channels, T).  Written as 32-bit float WAV (what torchaudio.save does for float32)."""
    ta = _torchaudio()
    if ta is not None:
        return ta.save(str(path), audio, fs)
    path = Path(path)
    if path.suffix.lower() != ".wav":
        raise RuntimeError(f"{path}: only .wav can be encoded without torchaudio")
    x = audio.detach().to(torch.float32).cpu().numpy()
    if x.ndim == 1:
        x = x[None]
    ch, n = x.shape
    payload = np.ascontiguousarray(x.T).astype("<f4").tobytes()
    fmt = (3).to_bytes(2, "little") + ch.to_bytes(2, "little") + int(fs).to_bytes(4, "little") + \
        (int(fs) * ch * 4).to_bytes(4, "little") + (ch * 4).to_bytes(2, "little") + (32).to_bytes(2, "little")
    fact = n.to_bytes(4, "little")
    body = b"WAVE" + b"fmt " + len(fmt).to_bytes(4, "little") + fmt + b"fact" + (4).to_bytes(4, "little") + fact + \
        b"data" + len(payload).to_bytes(4, "little") + payload
    with open(path, "wb") as f:
        f.write(b"RIFF" + len(body).to_bytes(4, "little") + body)


_kernel_cache = {}


def _sinc_kernel(orig, new, device, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional.resample's `sinc_interp_hann` kernel: (new, 1, 2*width + orig) and `width`."""
    key = (orig, new, str(device))
    if key not in _kernel_cache:
        base = min(orig, new) * rolloff
        width = math.ceil(lowpass_filter_width * orig / base)
        idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
        t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
        t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        t = t * math.pi
        scale = base / orig
        k = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * scale
        _kernel_cache[key] = (k.to(torch.float32).to(device), width)
    return _kernel_cache[key]


def resample(audio, fs, target_fs):
    """torchaudio.functional.resample(audio, fs, target_fs) with its default arguments (bin/enhance.py:77-80):
    polyphase windowed-sinc FIR, evaluated as one strided conv1d on the tensor's device."""
    fs, target_fs = int(fs), int(target_fs)
    if fs == target_fs:
        return audio
    g = math.gcd(fs, target_fs)
    orig, new = fs // g, target_fs // g
    kernel, width = _sinc_kernel(orig, new, audio.device)
    shape = audio.shape
    x = audio.reshape(-1, shape[-1]).to(torch.float32)
    n, length = x.shape
    x = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernel, stride=orig)  # (n, new, frames)
    y = y.transpose(1, 2).reshape(n, -1)
    target_length = int(math.ceil(new * length / orig))
    return y[..., :target_length].reshape(shape[:-1] + (target_length,))
