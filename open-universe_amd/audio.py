"""Audio file I/O and resampling for the `enhance` CLI (the caller side of the hot path).

The reference CLI uses `torchaudio.load / save / functional.resample` (bin/enhance.py:77-80,180-192).  torchaudio
is used when it is installed; otherwise .wav files are read / written with the standard library + numpy, .flac files are
decoded by the native library (`ou_flac_decode`, csrc/ou_flac.cpp: every CRC of the stream and the MD5 of the decoded audio
are verified), and `resample` is restated from torchaudio's documented algorithm (`sinc_interp_hann`, lowpass_filter_width 6,
rolloff 0.99 -- the defaults the reference relies on).  .mp3 needs torchaudio.
"""
import ctypes
import hashlib
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
    """True when `load` can read this file here: anything with torchaudio, .wav / .flac without."""
    return _torchaudio() is not None or Path(path).suffix.lower() in (".wav", ".flac")


def _flac_info(raw):
    from . import _lib

    L = _lib.load()
    buf = (ctypes.c_uint8 * len(raw)).from_buffer_copy(raw)
    fs, ch, bps, total = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    md5 = (ctypes.c_uint8 * 16)()
    rc = L.ou_flac_info(buf, len(raw), ctypes.byref(fs), ctypes.byref(ch), ctypes.byref(bps), ctypes.byref(total), md5)
    if rc != 0:
        raise RuntimeError((L.ou_flac_last_error() or b"FLAC decoding failed").decode())
    return L, buf, fs.value, ch.value, bps.value, total.value, bytes(md5)


def _flac_channels(path):
    """Channel count of a FLAC file from its STREAMINFO block alone: the metadata block headers are walked with seeks, so
    neither the audio frames nor large metadata blocks (cover art) are read."""
    with open(path, "rb") as f:
        head = f.read(10)
        off = 0
        if head[:3] == b"ID3" and len(head) == 10:  # ID3v2 tag in front of the stream: sync-safe size
            sz = ((head[6] & 0x7F) << 21) | ((head[7] & 0x7F) << 14) | ((head[8] & 0x7F) << 7) | (head[9] & 0x7F)
            off = 10 + sz + (10 if head[5] & 0x10 else 0)
        f.seek(off)
        if f.read(4) != b"fLaC":
            raise RuntimeError("not a FLAC stream (no fLaC marker)")
        while True:
            h = f.read(4)
            if len(h) < 4:
                raise RuntimeError("FLAC: truncated metadata")
            last, typ, ln = h[0] & 0x80, h[0] & 0x7F, int.from_bytes(h[1:4], "big")
            if typ == 0:
                s = f.read(ln)
                if len(s) < 34 or ln < 34:
                    raise RuntimeError("FLAC: short STREAMINFO")
                return ((s[12] >> 1) & 7) + 1
            f.seek(ln, 1)
            if last:
                raise RuntimeError("FLAC: no STREAMINFO block")


def load_flac(path):
    """-> (float32 tensor (channels, T) in [-1, 1), sample rate): what torchaudio.load returns for a FLAC file (integer samples
    scaled by 2^-(bits - 1)).  Refuses a stream whose CRCs or whose MD5 signature of the decoded audio do not match."""
    raw = Path(path).read_bytes()
    try:
        L, buf, fs, ch, bps, total, md5 = _flac_info(raw)
        # STREAMINFO is untrusted input (36 bits of sample count x up to 8 channels): a frame of a few bytes can stand for a
        # whole block of 65 535 constant samples per channel, nothing can stand for more
        if total > max(1, len(raw)) * 16384:
            raise RuntimeError(f"FLAC: STREAMINFO claims {total} samples per channel, more than {len(raw)} bytes can encode")
        try:
            out = np.zeros((ch, max(total, 1)), dtype=np.int32)
        except MemoryError:
            raise RuntimeError(f"FLAC: cannot hold {ch} x {total} samples in memory") from None
        done = ctypes.c_int64()
        rc = L.ou_flac_decode(buf, len(raw), out.ctypes.data_as(ctypes.c_void_p), out.shape[1], ctypes.byref(done))
        if rc != 0:
            raise RuntimeError((L.ou_flac_last_error() or b"FLAC decoding failed").decode())
    except RuntimeError as e:
        raise RuntimeError(f"{path}: {e}") from None
    out = out[:, :done.value]
    if any(md5):  # MD5 of the interleaved little-endian samples, ceil(bits / 8) bytes each
        nb = (bps + 7) // 8
        inter = np.ascontiguousarray(out.T).astype("<i4")
        b = inter.view(np.uint8).reshape(-1, 4)[:, :nb] if nb < 4 else inter.view(np.uint8).reshape(-1, 4)
        if hashlib.md5(np.ascontiguousarray(b).tobytes()).digest() != md5:
            raise RuntimeError(f"{path}: FLAC MD5 signature of the decoded audio does not match the stream's")
    return torch.from_numpy(out.astype(np.float32) / float(1 << (bps - 1))), fs


def load(path):
    """-> (float32 tensor (channels, T) in [-1, 1], sample rate): torchaudio.load semantics."""
    ta = _torchaudio()
    if ta is not None:
        return ta.load(str(path))
    path = Path(path)
    if path.suffix.lower() == ".flac":
        return load_flac(path)
    if path.suffix.lower() != ".wav":
        raise RuntimeError(f"{path}: only .wav and .flac can be decoded without torchaudio")
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
    if Path(path).suffix.lower() == ".flac":
        try:
            return _flac_channels(path)
        except RuntimeError as e:
            raise RuntimeError(f"{path}: {e}") from None
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
    if path.suffix.lower() == ".flac":
        return save_flac(path, audio, fs)
    if path.suffix.lower() != ".wav":
        raise RuntimeError(f"{path}: only .wav and .flac can be encoded without torchaudio")
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


# ---- FLAC encoder (the output side of a .flac input: the reference writes the enhanced file under the input's name with
# torchaudio.save, bin/enhance.py:77-80).  A plain, valid subset of the format, vectorised with numpy: independent channels,
# fixed block size, the order-2 fixed predictor, one Rice partition per subframe (verbatim where that is shorter), 24 bits per
# sample (float32 audio; 16 on request), STREAMINFO with the MD5 of the audio.  Compression ratio is not the point; any FLAC
# decoder reads the result, `load_flac` reads it back bit for bit (tests/test_audio_flac.py).
_CRC8 = None
_CRC16 = None


def _crc_tables():
    global _CRC8, _CRC16
    if _CRC8 is None:
        t8, t16 = [], []
        for i in range(256):
            c = i
            for _ in range(8):
                c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
            t8.append(c)
            c = i << 8
            for _ in range(8):
                c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
            t16.append(c)
        _CRC8, _CRC16 = t8, t16
    return _CRC8, _CRC16


def _crc8(b):
    t, _ = _crc_tables()
    c = 0
    for x in b:
        c = t[c ^ x]
    return c


def _crc16(b):
    _, t = _crc_tables()
    c = 0
    for x in b:
        c = ((c << 8) & 0xFFFF) ^ t[(c >> 8) ^ x]
    return c


def _bits_of(values, width):
    """(n,) non-negative ints -> (n, width) array of bits, MSB first"""
    v = np.asarray(values, dtype=np.uint64)[:, None]
    sh = np.arange(width - 1, -1, -1, dtype=np.uint64)[None, :]
    return ((v >> sh) & np.uint64(1)).astype(np.uint8)


def _utf8_number(n):
    if n < 0x80:
        return bytes([n])
    out, lead = [], 0
    nb = 2 if n < 0x800 else 3 if n < 0x10000 else 4 if n < 0x200000 else 5 if n < 0x4000000 else 6 if n < 0x80000000 else 7
    for _ in range(nb - 1):
        out.append(0x80 | (n & 0x3F))
        n >>= 6
    lead = ((0xFF << (8 - nb)) & 0xFF) | n
    return bytes([lead] + out[::-1])


def _subframe_bits(x, bps):
    """one channel of one block (int64) -> bit array: order-2 fixed predictor + one Rice partition, or verbatim"""
    n = len(x)
    two = np.uint64(1) << np.uint64(bps)
    verb = np.concatenate([np.array([0, 0, 0, 0, 0, 0, 1, 0], np.uint8), _bits_of((x.astype(np.int64) % int(two)).astype(np.uint64), bps).ravel()])
    if n < 3:
        return verb
    r = x[2:] - 2 * x[1:-1] + x[:-2]
    u = np.where(r >= 0, 2 * r, -2 * r - 1).astype(np.uint64)
    mean = float(u.mean()) if len(u) else 0.0
    k = int(max(0, min(14, math.floor(math.log2(mean + 1.0)))))
    q = (u >> np.uint64(k)).astype(np.int64)
    if int(q.max(initial=0)) > 4096:
        return verb
    lens = q + 1 + k
    total = int(lens.sum())
    if 8 + 2 * bps + 10 + total >= len(verb):
        return verb
    body = np.zeros(total, np.uint8)
    ends = np.cumsum(lens)
    body[ends - k - 1] = 1                                    # the terminating 1 of every unary part
    if k:
        low = _bits_of(u & np.uint64((1 << k) - 1), k)
        idx = (ends - k)[:, None] + np.arange(k)[None, :]
        body[idx.ravel()] = low.ravel()
    head = np.concatenate([np.array([0, 0, 0, 1, 0, 1, 0, 0], np.uint8),               # pad, fixed order 2, no wasted bits
                           _bits_of((x[:2].astype(np.int64) % int(two)).astype(np.uint64), bps).ravel(),
                           np.array([0, 0], np.uint8), np.zeros(4, np.uint8), _bits_of([k], 4).ravel()])
    return np.concatenate([head, body])


def flac_encode(samples, fs, bps=24, block=4096):
    """samples: int array (channels, T), values within bps bits -> the bytes of a FLAC stream."""
    x = np.asarray(samples, dtype=np.int64)
    ch, n = x.shape
    if not (1 <= ch <= 8) or bps not in (8, 16, 24) or not (0 < fs < (1 << 20)):
        raise ValueError("flac_encode: 1-8 channels, 8 / 16 / 24 bits per sample")
    nb = (bps + 7) // 8
    inter = np.ascontiguousarray(x.T).astype("<i4").view(np.uint8).reshape(-1, 4)[:, :nb]
    md5 = hashlib.md5(np.ascontiguousarray(inter).tobytes()).digest()
    frames, fno, sizes = [], 0, []
    ssc = {8: 1, 16: 4, 24: 6}[bps]
    for s0 in range(0, max(n, 1), block):
        blk = x[:, s0:s0 + block]
        bs = blk.shape[1]
        if bs == 0:
            break
        hdr = bytearray([0xFF, 0xF8, (7 << 4) | 0, ((ch - 1) << 4) | (ssc << 1)])   # 16-bit block size follows, rate from STREAMINFO
        hdr += _utf8_number(fno)
        hdr += (bs - 1).to_bytes(2, "big")
        hdr.append(_crc8(hdr))
        bits = np.concatenate([_subframe_bits(blk[c], bps) for c in range(ch)])
        body = np.packbits(bits).tobytes()
        fr = bytes(hdr) + body
        fr += _crc16(fr).to_bytes(2, "big")
        frames.append(fr)
        sizes.append(len(fr))
        fno += 1
    si = bytearray()
    si += min(block, 65535).to_bytes(2, "big") * 2
    si += (min(sizes) if sizes else 0).to_bytes(3, "big") + (max(sizes) if sizes else 0).to_bytes(3, "big")
    v = (int(fs) << 44) | ((ch - 1) << 41) | ((bps - 1) << 36) | n
    si += v.to_bytes(8, "big") + md5
    return b"fLaC" + bytes([0x80]) + len(si).to_bytes(3, "big") + bytes(si) + b"".join(frames)


def save_flac(path, audio, fs, bits_per_sample=24):
    """audio: float tensor (channels, T) in [-1, 1] -> FLAC with `bits_per_sample` bits (round to nearest, clipped)."""
    x = audio.detach().to(torch.float64).cpu().numpy()
    if x.ndim == 1:
        x = x[None]
    full = float(1 << (bits_per_sample - 1))
    q = np.clip(np.rint(x * full), -full, full - 1).astype(np.int64)
    Path(path).write_bytes(flac_encode(q, int(fs), bits_per_sample))
