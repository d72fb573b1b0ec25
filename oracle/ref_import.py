"""
TEST INFRASTRUCTURE -- not part of the product path.

Import harness for the *real* reference (`/root/reference`, read-only, this container only).

The reference's hot path (`open_universe.networks.universe`) is plain PyTorch, but it imports framework
packages that are not installed here (hydra, omegaconf, pytorch_lightning, torch_ema, torchaudio).
This module registers minimal stand-ins for those *framework* packages in `sys.modules` (none of them
carries arithmetic of the hot path, except torchaudio's MelSpectrogram / Resample, which are restated
from torchaudio's documented algorithm -- parity at that third-party boundary is therefore UNPINNED,
see DESIGN.md) and then imports the reference's own, unmodified network code from where it lies.

Used only by `tests/golden/make_golden.py` (fixture generation) and by the CPU tests that validate
`oracle/restatement.py` against the reference when `/root/reference` is present.  Nothing here travels
to, or is used on, the GPU box.
"""
import importlib
import math
import os
import sys
import types

import torch
import yaml

REFERENCE_ROOT = os.environ.get("OU_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "open_universe", "networks", "universe"))


class AttrDict(dict):
    """dict with attribute access; the reference uses both cfg.key and cfg.get(key)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def wrap(obj):
    if isinstance(obj, dict):
        return AttrDict({k: wrap(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [wrap(v) for v in obj]
    return obj


def _instantiate(cfg, *args, _recursive_=False, _convert_=None, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    mod_name, cls_name = target.rsplit(".", 1)
    cls = getattr(importlib.import_module(mod_name), cls_name)
    cfg.update(kwargs)
    cfg = {k: wrap(v) for k, v in cfg.items()}
    return cls(*args, **cfg)


# --------------------------------------------------------------------------------------------------
# torchaudio stand-ins (documented algorithm of torchaudio.transforms.MelSpectrogram / Resample)
# --------------------------------------------------------------------------------------------------
def melscale_fbanks_htk(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk')."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down_slopes, up_slopes))
    return fb


class _Spectrogram(torch.nn.Module):
    def __init__(self, n_fft, hop_length):
        super().__init__()
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.register_buffer("window", torch.hann_window(n_fft), persistent=True)

    def forward(self, x):
        shape = x.shape
        x = x.reshape(-1, shape[-1])
        spec = torch.stft(
            x,
            self.n_fft,
            self.hop_length,
            self.n_fft,
            self.window,
            center=False,
            onesided=True,
            normalized=False,
            return_complex=True,
        )
        spec = spec.reshape(shape[:-1] + spec.shape[-2:])
        return spec.abs().pow(2.0)


class _MelScale(torch.nn.Module):
    def __init__(self, n_mels, sample_rate, n_stft):
        super().__init__()
        fb = melscale_fbanks_htk(n_stft, 0.0, float(sample_rate // 2), n_mels, sample_rate)
        self.register_buffer("fb", fb, persistent=True)

    def forward(self, spec):
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)


class MelSpectrogram(torch.nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, hop_length=None, n_mels=128, center=True, **kw):
        super().__init__()
        assert center is False
        self.spectrogram = _Spectrogram(n_fft, hop_length)
        self.mel_scale = _MelScale(n_mels, sample_rate, n_fft // 2 + 1)

    def forward(self, x):
        return self.mel_scale(self.spectrogram(x))


def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional._get_sinc_resample_kernel, 'sinc_interp_hann'."""
    gcd = math.gcd(int(orig_freq), int(new_freq))
    orig_freq = int(orig_freq) // gcd
    new_freq = int(new_freq) // gcd
    base_freq = min(orig_freq, new_freq) * rolloff
    width = math.ceil(lowpass_filter_width * orig_freq / base_freq)
    idx = torch.arange(-width, width + orig_freq, dtype=torch.float64)[None, None] / orig_freq
    t = torch.arange(0, -new_freq, -1, dtype=torch.float64)[:, None, None] / new_freq + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig_freq
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32), width, orig_freq, new_freq


def apply_sinc_resample_kernel(x, orig_freq, new_freq, kernel, width):
    shape = x.shape
    x = x.reshape(-1, shape[-1])
    length = x.shape[-1]
    x = torch.nn.functional.pad(x, (width, width + orig_freq))
    res = torch.nn.functional.conv1d(x[:, None], kernel, stride=orig_freq)
    res = res.transpose(1, 2).reshape(x.shape[0], -1)
    target_length = int(math.ceil(new_freq * length / orig_freq))
    res = res[..., :target_length]
    return res.reshape(shape[:-1] + res.shape[-1:])


class Resample(torch.nn.Module):
    def __init__(self, orig_freq=16000, new_freq=16000, **kw):
        super().__init__()
        kernel, self.width, self.orig_freq, self.new_freq = sinc_resample_kernel(orig_freq, new_freq)
        self.register_buffer("kernel", kernel, persistent=True)

    def forward(self, x):
        if self.orig_freq == self.new_freq:
            return x
        return apply_sinc_resample_kernel(x, self.orig_freq, self.new_freq, self.kernel, self.width)


class ExponentialMovingAverage:
    """torch_ema stand-in: state layout only (shadow_params list in parameter order)."""

    def __init__(self, parameters, decay):
        parameters = list(parameters)
        self.decay = decay
        self.num_updates = 0
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.collected_params = None

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, parameters):
            p.data.copy_(s.data)

    def restore(self, parameters):
        for c, p in zip(self.collected_params, parameters):
            p.data.copy_(c.data)

    def to(self, *a, **k):
        self.shadow_params = [p.to(*a, **k) for p in self.shadow_params]

    def state_dict(self):
        return {
            "decay": self.decay,
            "num_updates": self.num_updates,
            "shadow_params": self.shadow_params,
            "collected_params": self.collected_params,
        }

    def load_state_dict(self, sd):
        self.decay = sd["decay"]
        self.num_updates = sd["num_updates"]
        self.shadow_params = [p.clone() for p in sd["shadow_params"]]


_INSTALLED = False


def install_stubs():
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    hu = mod("hydra.utils", instantiate=_instantiate, to_absolute_path=lambda p: p)
    mod("hydra", utils=hu)

    class OmegaConf:
        @staticmethod
        def create(d):
            return wrap(d)

        @staticmethod
        def to_container(d, resolve=True):
            return dict(d)

    mod("omegaconf", OmegaConf=OmegaConf, DictConfig=AttrDict)

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

        def log(self, *a, **k):
            pass

    mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=object)
    mod("torch_ema", ExponentialMovingAverage=ExponentialMovingAverage)
    ta_t = mod("torchaudio.transforms", MelSpectrogram=MelSpectrogram, Resample=Resample)
    ta_f = mod("torchaudio.functional")
    mod("torchaudio", transforms=ta_t, functional=ta_f)

    pkg = types.ModuleType("open_universe")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "open_universe")]
    sys.modules["open_universe"] = pkg
    _INSTALLED = True


def _coerce(v):
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            return v
    return v


def load_reference_config(name, overrides=None):
    """Read config/model/<name>.yaml of the reference and resolve ${...} interpolations."""
    root = os.path.join(REFERENCE_ROOT, "config")
    with open(os.path.join(root, "model", name + ".yaml")) as f:
        model = yaml.safe_load(f)
    with open(os.path.join(root, "datamodule", "default.yaml")) as f:
        datamodule = yaml.safe_load(f)
    with open(os.path.join(root, "trainer", "default.yaml")) as f:
        trainer = yaml.safe_load(f)
    tree = {"model": model, "datamodule": datamodule, "trainer": trainer}

    def lookup(path):
        node = tree
        for p in path.split("."):
            if not isinstance(node, dict) or p not in node:
                return 1.0  # unresolved (training-only) interpolation
            node = node[p]
        return resolve(node)

    def resolve(node):
        if isinstance(node, dict):
            return {k: resolve(v) for k, v in node.items()}
        if isinstance(node, list):
            return [resolve(v) for v in node]
        if isinstance(node, str) and node.startswith("${") and node.endswith("}"):
            return lookup(node[2:-1])
        return _coerce(node)

    model = resolve(model)
    model["validation"]["enh_losses"] = {}
    for key, val in (overrides or {}).items():
        node = model
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = val
    return model


def build_reference_model(name, overrides=None):
    """Instantiate the reference's Universe / UniverseGAN from its own yaml config (CPU)."""
    install_stubs()
    importlib.import_module("open_universe.networks.universe")
    cfg = load_reference_config(name, overrides)
    model = _instantiate(cfg, _recursive_=False)
    return model, cfg
