"""
Signal pre-conditioning transforms of the reference (open_universe/layers/dyn_range_comp.py:28-225) on the HIP kernels
of libouniverse.so: same classes, constructor arguments, call conventions and error behaviour
(`transform(x)`, `transform(x, inv=True, length=...)`, `transform.inv(x)`; input (batch, 1, time), output
(batch, 2 * n_freq, frames) with the real parts stacked before the imaginary parts).

No shipped model config sets `transform` (`universe.py:112-115`): the UNIVERSE / UNIVERSE++ checkpoints run on the
identity.  The STFT-domain variants are here so that a config that does name them loads the same arithmetic.
"""
import ctypes
from ctypes import c_void_p

import torch

from .. import _lib

_TYPES = {"none": 0, "exponent": 1, "log": 2}


class IdentityTransform:
    """dyn_range_comp.py:28-37."""

    def __call__(self, x, inv=None):
        return x

    def inv(self, x):
        return self(x)


def get_window(window_type, window_length):
    """dyn_range_comp.py:40-48."""
    if window_type == "sqrthann":
        return torch.sqrt(torch.hann_window(window_length, periodic=True))
    if window_type == "hann":
        return torch.hann_window(window_length, periodic=True)
    if window_type == "hamming":
        return torch.hamming_window(window_length, periodic=True)
    raise NotImplementedError(f"Window type {window_type} not implemented!")


class CompressedMagSTFT:
    """dyn_range_comp.py:51-170."""

    def __init__(self, stft_kwargs, spec_kwargs, inv=False):
        assert all(k in stft_kwargs for k in ["n_fft", "hop_length", "window_name"])
        assert all(k in spec_kwargs for k in ["transform_type", "abs_exponent", "factor"])
        extra = set(stft_kwargs) - {"n_fft", "hop_length", "window_name"}
        if extra:
            raise NotImplementedError(f"stft_kwargs {sorted(extra)} are not supported (win_length = n_fft only)")
        if spec_kwargs["transform_type"] not in _TYPES:
            raise NotImplementedError(f"transform_type {spec_kwargs['transform_type']!r}")
        self._inv = inv
        self.n_fft = int(stft_kwargs["n_fft"])
        self.hop_length = int(stft_kwargs["hop_length"])
        self.transform_type = spec_kwargs["transform_type"]
        self.abs_exponent = float(spec_kwargs["abs_exponent"])
        self.factor = float(spec_kwargs["factor"])
        self.stft_window = get_window(stft_kwargs.get("window_name", "hann"), self.n_fft)
        self._L = _lib.load()  # no fallback: raises when the extension is not built

    # ---- device plumbing ------------------------------------------------------------------------------------------
    def _window_on(self, device):
        if self.stft_window.device != device:
            self.stft_window = self.stft_window.to(device)
        return self.stft_window

    def _args(self):
        return _TYPES[self.transform_type], self.abs_exponent, self.factor

    @staticmethod
    def _stream(device):
        return c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def _need_gpu(self, x):
        if x.device.type != "cuda":
            raise RuntimeError("open_universe_amd transforms run on a HIP device only (no CPU path)")
        return x.to(torch.float32).contiguous()

    # ---- reference interface --------------------------------------------------------------------------------------
    def __call__(self, x, inv=False, length=None):
        return self.forward(x, inv=inv, length=length)

    def forward(self, x, inv=False, length=None):
        if self._inv:
            inv = not inv
        if not inv:
            if x.shape[1] != 1:  # (checked before the rank, like the reference, dyn_range_comp.py:75-78)
                raise ValueError("Expects single channel input")
            if x.ndim != 3:
                raise ValueError("Expects a 3D input tensor (batch, channels, time)")
            return self._stft(self._need_gpu(x).squeeze(1))
        if x.ndim != 3:
            raise ValueError("Expects a 3D input tensor (batch, freq/real/imag, time)")
        return self._istft(self._need_gpu(x), length=length).unsqueeze(1)

    def inv(self, x, length=None):
        return self(x, inv=True, length=length)

    def _stft(self, sig):
        B, T = sig.shape  # (any length >= 1: the centre padding is zeros -- pad_mode="constant" --, not a reflection)
        F = self.n_fft // 2 + 1
        n_frames = self._L.ou_transform_frames(T, self.n_fft, self.hop_length)
        out = torch.empty(B, 2 * F, n_frames, dtype=torch.float32, device=sig.device)
        t, e, f = self._args()
        with torch.cuda.device(sig.device):
            _lib.check(self._L.ou_transform_forward(c_void_p(sig.data_ptr()), B, T,
                                                    c_void_p(self._window_on(sig.device).data_ptr()), self.n_fft,
                                                    self.hop_length, t, e, f, c_void_p(out.data_ptr()),
                                                    self._stream(sig.device)))
        return out

    def _istft(self, spec, length=None):
        B, C2, n_frames = spec.shape
        if C2 != 2 * (self.n_fft // 2 + 1):
            raise ValueError(f"expected {2 * (self.n_fft // 2 + 1)} channels (real | imag of {self.n_fft // 2 + 1} bins)")
        if length is None:
            length = self.hop_length * (n_frames - 1)  # torch.istft default
        y = torch.empty(B, length, dtype=torch.float32, device=spec.device)
        scratch = torch.empty(B * n_frames * self.n_fft, dtype=torch.float32, device=spec.device)
        t, e, f = self._args()
        with torch.cuda.device(spec.device):
            _lib.check(self._L.ou_transform_inverse(c_void_p(spec.data_ptr()), B, n_frames,
                                                    c_void_p(self._window_on(spec.device).data_ptr()), self.n_fft,
                                                    self.hop_length, t, e, f, int(length), c_void_p(y.data_ptr()),
                                                    c_void_p(scratch.data_ptr()), self._stream(spec.device)))
        return y


class CompressedMagSTFTPadded(CompressedMagSTFT):
    """dyn_range_comp.py:173-225: pads to a multiple of `pad_block` and drops the last hop so that the centred STFT has
    a whole number of blocks of frames.  The reference applies `_pad` TWICE in `_stft` (:199-201); so does this."""

    def __init__(self, stft_kwargs, spec_kwargs, pad_block=None, inv=False):
        super().__init__(stft_kwargs, spec_kwargs, inv=inv)
        if pad_block is not None:
            if pad_block % self.hop_length != 0:
                raise ValueError("pad_block must be a multiple of hop_length")
            self.pad_block = pad_block
        else:
            self.pad_block = 0

    def _pad(self, x):
        if self.pad_block > 0:
            r = x.shape[-1] % self.pad_block
            if r > 0:
                x = torch.nn.functional.pad(x, (0, self.pad_block - r), mode="constant", value=0.0)
        return x[..., : -self.hop_length]

    def _stft(self, sig):
        return super()._stft(self._pad(self._pad(sig)).contiguous())

    def _istft(self, spec, length=None):
        if length is None:
            length = spec.shape[-1] * self.hop_length
        return super()._istft(spec, length=length)
