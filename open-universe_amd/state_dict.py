"""
State-dict key schema of the reference networks, synthetic (seeded) weights, and EMA mapping.

The reference stores weights in a Lightning checkpoint: `{"state_dict": {...}, "ema": {...}}`
(open_universe/inference_utils/model_loader.py:117-130; networks/universe/universe.py:832-869).
This module re-derives, from a `ModelSpec`, the ordered list of tensors the reference registers
(name, shape, parameter|buffer) -- needed to
  * map `ema["shadow_params"]` (a list in `model_parameters()` order, universe.py:130-133,
    universe_gan.py:136-143) back onto names,
  * generate seeded synthetic checkpoints (no trained checkpoint is reachable offline),
  * validate a checkpoint before it is handed to the C ABI.
The schema is checked against the imported reference in tests/test_oracle_vs_reference.py and against
the committed key lists in tests/golden/.
"""
import math
from typing import Dict, List, Tuple

import torch

from .config import ModelSpec, NetSpec

Entry = Tuple[str, Tuple[int, ...], bool]  # (key, shape, is_parameter)


def _conv(p, cout, cin, k, wn, bias=True, transpose=False) -> List[Entry]:
    shape = (cin, cout, k) if transpose else (cout, cin, k)
    if wn:
        out = [(p + ".bias", (cout,), True)] if bias else []
        return out + [(p + ".weight_g", (shape[0], 1, 1), True), (p + ".weight_v", shape, True)]
    out = [(p + ".weight", shape, True)]
    return out + ([(p + ".bias", (cout,), True)] if bias else [])


def _linear(p, cout, cin, wn) -> List[Entry]:
    if wn:
        return [(p + ".bias", (cout,), True), (p + ".weight_g", (cout, 1), True), (p + ".weight_v", (cout, cin), True)]
    return [(p + ".weight", (cout, cin), True), (p + ".bias", (cout,), True)]


def _prelu_conv(p, cin, cout, k, wn, aa=False, transpose=False) -> List[Entry]:
    """blocks.py:133-203 registration order: [bias, low_pass_filter.weights], prelu, conv."""
    out: List[Entry] = []
    if aa:
        out += [(p + ".bias", (cout,), True), (p + ".low_pass_filter.weights", (2 * k + 1,), False)]
    out += [(p + ".prelu.weight", (1,), True)]
    out += _conv(p + ".conv", cout, cin, k, wn, bias=not aa, transpose=transpose)
    return out


def _conv_block(p, c, wn, rate=None, direction="none", aa=False) -> List[Entry]:
    """blocks.py:236-312: rate_change_conv is registered before conv1..3."""
    out: List[Entry] = []
    if direction == "down":
        out += _prelu_conv(p + ".rate_change_conv", c, 2 * c, rate, wn, aa)
    elif direction == "up":
        out += _prelu_conv(p + ".rate_change_conv", 2 * c, c, rate, wn, aa, transpose=True)
    for name, k in (("conv1", 5), ("conv2", 3), ("conv3", 3)):
        out += _prelu_conv(f"{p}.{name}", c, c, k, wn)
    return out


def _gru(p, inp, hid, layers) -> List[Entry]:
    out: List[Entry] = []
    for layer in range(layers):
        i = inp if layer == 0 else 2 * hid
        for sfx in ("", "_reverse"):
            k = f"_l{layer}{sfx}"
            out += [(p + ".weight_ih" + k, (3 * hid, i), True), (p + ".weight_hh" + k, (3 * hid, hid), True),
                    (p + ".bias_ih" + k, (3 * hid,), True), (p + ".bias_hh" + k, (3 * hid,), True)]
    return out


def score_schema(p: str, s: NetSpec) -> List[Entry]:
    """score.py:213-273 ScoreNetwork (+ encoder :26-102, decoder :130-194)."""
    c0, rates, wn, aa, D = s.n_channels, s.rate_factors, s.use_weight_norm, s.use_antialiasing, s.noise_cond_dim
    out: List[Entry] = []
    if s.time_embedding == "simple":
        out += [(p + ".sigma_block.weight", (1, 1), True), (p + ".sigma_block.bias", (1, 1), True)]
    else:
        out += [(p + ".sigma_block.freq", (s.n_rff,), False)]
        dims = [2 * s.n_rff, 4 * s.n_rff, 8 * s.n_rff, D]
        for i in range(3):
            q = f"{p}.sigma_block.layer{i + 1}"
            out += [(q + ".prelu.weight", (1,), True), (q + ".lin.weight", (dims[i + 1], dims[i]), True),
                    (q + ".lin.bias", (dims[i + 1],), True)]
    out += _conv(p + ".input_conv", c0, 1, s.fb_kernel_size, False)
    n = len(rates)
    oc = c0 * 2 ** n
    for i, r in enumerate(rates):
        out += _conv_block(f"{p}.encoder.ds_modules.{i}", c0 * 2 ** i, wn, r, "down", aa)
    if s.extra_conv_block:
        out += _conv_block(f"{p}.encoder.ds_modules.{n}", oc, wn)
    for i in range(n):
        out += _linear(f"{p}.encoder.cond_proj.{i}", c0 * 2 ** (i + 1), D, wn)
    if s.extra_conv_block:
        out += _linear(f"{p}.encoder.cond_proj.{n}", 2 * oc, D, wn)
    out += _gru(p + ".encoder.gru", oc, oc // 2, 1)
    chans = [c0 * 2 ** (n - i - 1) for i in range(n)]
    up = rates[::-1]
    blocks = ([(oc, None, "none")] if s.extra_conv_block else []) + [(c, r, "up") for c, r in zip(chans, up)]
    for j, (c, r, d) in enumerate(blocks):
        out += _conv_block(f"{p}.decoder.up_modules.{j}", c, wn, r, d, aa)
    for j, (c, r, d) in enumerate(blocks):
        out += _linear(f"{p}.decoder.noise_cond_proj.{j}", 2 * c, D, wn)
    for j, (c, r, d) in enumerate(blocks):
        out += _conv(f"{p}.decoder.signal_cond_proj.{j}", c, c, 1, wn)
    out += [(p + ".prelu.weight", (1,), True)]
    out += _prelu_conv(p + ".output_conv", c0, 1, s.fb_kernel_size, wn)
    return out


def cond_schema(p: str, c: NetSpec) -> List[Entry]:
    """condition.py:273-342 ConditionerNetwork (MelAdapter :68-90, encoder :117-187, decoder :223-262)."""
    c0, rates, wn = c.n_channels, c.rate_factors, c.use_weight_norm
    n = len(rates)
    oc = c0 * 2 ** n
    hop = math.prod(rates)
    n_fft = c.n_mel_oversample * hop
    out: List[Entry] = _conv(p + ".input_conv", c0, 1, c.fb_kernel_size, wn)
    out += [(p + ".input_mel.mel_spec.spectrogram.window", (n_fft,), False),
            (p + ".input_mel.mel_spec.mel_scale.fb", (n_fft // 2 + 1, c.n_mels), False)]
    out += _conv(p + ".input_mel.conv", oc, c.n_mels, 3, wn)
    out += _conv_block(p + ".input_mel.conv_block", oc, wn)
    for i, r in enumerate(rates):  # encoder never uses anti-aliasing (condition.py:333)
        out += _conv_block(f"{p}.encoder.ds_modules.{i}", c0 * 2 ** i, wn, r, "down", False)
    if c.extra_conv_block:
        out += _conv_block(f"{p}.encoder.ds_modules.{n}", oc, wn)
    for i in range(n - 1):  # make_st_convs condition.py:33-65
        out += _prelu_conv(f"{p}.encoder.st_convs.{i}", c0 * 2 ** i, oc, math.prod(rates[i:]), wn)
    out += _gru(p + ".encoder.gru", oc, oc // 2, 2)
    out += _conv_block(p + ".encoder.conv_block1", oc, wn)
    out += _conv_block(p + ".encoder.conv_block2", oc, wn)
    out += _conv_block(p + ".decoder.input_conv_block", oc, wn)
    chans = [c0 * 2 ** (n - i - 1) for i in range(n)]
    blocks = ([(oc, None, "none")] if c.extra_conv_block else []) + [(ch, r, "up") for ch, r in zip(chans, rates[::-1])]
    for j, (ch, r, d) in enumerate(blocks):
        out += _conv_block(f"{p}.decoder.up_modules.{j}", ch, wn, r, d, c.use_antialiasing)
    return out


def decoupling_schema(spec: ModelSpec) -> List[Entry]:
    """universe_gan.py:117-126: PReLU_Conv(n_channels -> 1, k=3, act=snake), no weight-norm."""
    if not spec.use_signal_decoupling:
        return []
    p, c0 = "signal_decoupling_layer", spec.score.n_channels
    out: List[Entry] = []
    if spec.signal_decoupling_act == "snake":
        out += [(p + ".prelu.act.act.alpha", (c0,), True), (p + ".prelu.act.upsample.kernel", (2, 1, 15), False),
                (p + ".prelu.act.downsample.kernel", (1, 1, 28), False)]
    elif spec.signal_decoupling_act in ("prelu", None):
        if spec.signal_decoupling_act == "prelu":
            out += [(p + ".prelu.weight", (1,), True)]
    else:
        raise NotImplementedError(f"signal_decoupling_act={spec.signal_decoupling_act}")
    return out + _conv(p + ".conv", 1, c0, 3, False)


def model_schema(spec: ModelSpec) -> List[Entry]:
    """All inference tensors in the reference's registration order (loss/discriminator modules excluded)."""
    return score_schema(spec.score_prefix, spec.score) + cond_schema("condition_model", spec.cond) + decoupling_schema(spec)


def parameter_names(spec: ModelSpec) -> List[str]:
    """`model_parameters()` order = EMA shadow_params order."""
    return [k for k, _, is_p in model_schema(spec) if is_p]


# --------------------------------------------------------------------------------------------------
# buffers (values a real checkpoint carries; recomputed here for synthetic checkpoints)
# --------------------------------------------------------------------------------------------------
def binomial_taps(kernel_size: int) -> torch.Tensor:
    """Reference blocks.py:62-68: binomial row scaled to unit RMS."""
    full = torch.zeros(kernel_size, kernel_size, dtype=torch.float64)
    for n in range(kernel_size):
        for i in range(n + 1):
            full[n, i] = math.comb(n, i)
    w = (full[kernel_size - 1] / full.square().mean().sqrt()).to(torch.float32)
    return w / w.square().mean().sqrt()


def mel_filterbank(n_freqs: int, n_mels: int, sample_rate: int = 24000) -> torch.Tensor:
    """torchaudio melscale_fbanks(htk, norm=None) with f_max = sample_rate//2; the reference hard-codes
    sample_rate=24000 (condition.py:75-81)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_max = 2595.0 * math.log10(1.0 + (float(sample_rate // 2) / 700.0))
    m_pts = torch.linspace(0.0, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def sinc_resample_kernel(orig: int, new: int, width_param: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """torchaudio sinc_interp_hann resampling kernel (alias_free_act.py:21-22 uses 1->2 and 2->1)."""
    g = math.gcd(orig, new)
    orig, new = orig // g, new // g
    base = min(orig, new) * rolloff
    width = math.ceil(width_param * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t *= base
    t = t.clamp_(-width_param, width_param)
    window = torch.cos(t * math.pi / width_param / 2) ** 2
    t *= math.pi
    k = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    k *= window * (base / orig)
    return k.to(torch.float32)


def buffer_value(key: str, shape) -> torch.Tensor:
    if key.endswith("low_pass_filter.weights"):
        return binomial_taps(shape[0])
    if key.endswith("spectrogram.window"):
        return torch.hann_window(shape[0])
    if key.endswith("mel_scale.fb"):
        return mel_filterbank(shape[0], shape[1])
    if key.endswith("upsample.kernel"):
        return sinc_resample_kernel(1, 2)
    if key.endswith("downsample.kernel"):
        return sinc_resample_kernel(2, 1)
    raise KeyError(key)


def synthetic_state_dict(spec: ModelSpec, seed: int = 0, gain: float = 1.0, tail: float = 0.0) -> Dict[str, torch.Tensor]:
    """Seeded random weights with the reference key schema, scaled so that activations stay O(1) through
    the whole stack (a random net with default init collapses to ~0 and would make parity trivial).
    Deterministic given (spec, seed): a recipe, not data.  `gain` > 1 gives the "stress" set; `tail` > 0 makes the
    per-channel weight-norm gains heavy-tailed (log-normal factor exp(tail * N(0, 1)) from a generator of its own, so the
    other draws of a seed do not move): a few output channels several times louder than the rest, as trained nets have."""
    gen = torch.Generator().manual_seed(seed)
    gen_tail = torch.Generator().manual_seed(7919 + seed)
    sd: Dict[str, torch.Tensor] = {}

    def rn(*shape):
        return torch.randn(*shape, generator=gen)

    schema = model_schema(spec)
    aa_convs = {k.replace("low_pass_filter.weights", "conv.weight_g") for k, _, _ in schema
                if k.endswith("low_pass_filter.weights")}
    for key, shape, is_p in schema:
        leaf = key.rsplit(".", 1)[-1]
        if not is_p:
            if leaf == "freq":
                v = 16.0 * rn(*shape)  # sigma_block.py:45
            else:
                v = buffer_value(key, shape)
        elif leaf == "weight_v":
            v = 0.05 * rn(*shape)
        elif leaf == "weight_g":
            v = gain * (1.25 + 0.1 * rn(*shape)).abs()
            if tail > 0.0:
                v = v * torch.exp(tail * torch.randn(*shape, generator=gen_tail) - 0.5 * tail * tail)
            if key in aa_convs:
                v = v / 4.0  # the unit-RMS binomial FIR has a DC gain of 4..8
        elif "prelu.weight" in key or key.endswith(".prelu.weight"):
            v = 0.25 + 0.05 * rn(*shape)
        elif leaf == "alpha":
            v = 0.3 * rn(*shape)
        elif key.endswith("sigma_block.weight"):
            v = torch.full(shape, 0.9)
        elif key.endswith("sigma_block.bias"):
            v = torch.full(shape, 0.2)
        elif "gru" in key:
            hid = shape[0] // 3
            v = (torch.rand(*shape, generator=gen) * 2 - 1) * (gain / math.sqrt(hid))
        elif leaf == "bias":
            v = 0.05 * rn(*shape)
            if "cond_proj" in key and "signal" not in key:  # FiLM: gamma ~ 1
                v[: shape[0] // 2] += 1.0
        elif leaf == "weight":  # un-normalised conv / linear: variance preserving
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            v = rn(*shape) * (gain * 1.25 / math.sqrt(fan_in))
            if len(shape) == 3 and "rate_change_conv" in key and "decoder" in key:
                v = v * math.sqrt(shape[2])  # ConvTranspose weight is (in, out, k): one tap per output
        else:
            raise KeyError(key)
        sd[key] = v.to(torch.float32).contiguous()
    return sd


def checkpoint_from_state_dict(spec: ModelSpec, sd, with_ema=True, ema_jitter=0.0, seed=1):
    """Lightning-style checkpoint dict.  With EMA, `shadow_params` holds the inference weights
    (universe.py:841-865: eval() copies the EMA weights over the parameters)."""
    ckpt = {"state_dict": dict(sd)}
    if with_ema and spec.ema_decay > 0.0:
        names = parameter_names(spec)
        gen = torch.Generator().manual_seed(seed)
        shadow = [sd[n] + ema_jitter * torch.randn(sd[n].shape, generator=gen) for n in names]
        ckpt["ema"] = {"decay": spec.ema_decay, "num_updates": 1, "shadow_params": shadow, "collected_params": None}
    return ckpt


def merge_lora(sd, lora_alpha=None):
    """Fold LoRA adapters back into plain weights (open_universe/lora/lora.py: `_get_weights` of LoraConv1d :68-71,
    LoraConvTranspose1d :144-147, LoraLinear :245-247) and strip the `model.` prefix a `UniverseLoRA` wrapper adds
    (networks/universe/lora.py:66-80 keeps the base model as `self.model`).

    `lora.remove()` (lora/utils.py:72-89) only un-wraps LoraConv1d / LoraLinear, so a "merged" checkpoint still carries
    adapters on the ConvTranspose1d layers; both forms -- and fully un-merged ones -- come out of here as the plain
    `<prefix>.weight` / `<prefix>.bias` tensors of the base model.  W = W0 + (alpha / rank) * (A @ B).view_as(W0);
    alpha is not stored in the checkpoint (default: alpha = rank -> scale 1)."""
    sd = dict(sd)
    if sd and all(k.startswith("model.") for k in sd):
        sd = {k[len("model."):]: v for k, v in sd.items()}
    for kind, inner in (("lora_weight", "conv"), ("lora_linear", "linear")):
        for ka in [k for k in sd if k.endswith("." + kind + "_a")]:
            pfx = ka[: -len(kind) - 3]
            a, b = sd.pop(ka), sd.pop(pfx + "." + kind + "_b")
            w0 = sd.pop(pfx + "." + inner + ".weight")
            rank = a.shape[1]
            scale = 1.0 if lora_alpha is None else float(lora_alpha) / rank
            sd[pfx + ".weight"] = w0 + scale * (a.double() @ b.double()).view(w0.shape).to(w0.dtype)
            if pfx + "." + inner + ".bias" in sd:
                sd[pfx + ".bias"] = sd.pop(pfx + "." + inner + ".bias")
    return sd


def inference_state_dict(spec: ModelSpec, ckpt, lora_alpha=None) -> Dict[str, torch.Tensor]:
    """Resolve the tensors inference runs on (model_loader.py:117-132 + universe.py:841-865):
    state_dict (loss/discriminator keys ignored), overwritten by the EMA shadow weights.

    Also accepts what fine-tuning leaves behind: LoRA adapters (merged here, see merge_lora) and
    weight-norm already removed (`Universe.remove_weight_norm`, universe.py:135-136 / blocks.py:45-50, called by
    UniverseLoRA._fix_model): a plain `<p>.weight` stands in for `<p>.weight_g` + `<p>.weight_v` -- the packer takes
    either.  Missing inference tensors are always fatal (the reference's strict=False would silently keep random
    initialisation there)."""
    has_wrapper = isinstance(ckpt, dict) and "state_dict" in ckpt
    sd_in = ckpt["state_dict"] if has_wrapper else ckpt
    had_lora = any(".lora_weight_" in k or ".lora_linear_" in k for k in sd_in)
    sd_in = merge_lora(sd_in, lora_alpha)
    schema = model_schema(spec)
    out: Dict[str, torch.Tensor] = {}
    missing = []
    unnormed = False
    for key, shape, is_p in schema:
        if key in sd_in:
            t = sd_in[key]
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"checkpoint tensor {key} has shape {tuple(t.shape)}, expected {shape}")
            out[key] = t
        elif key.endswith(".weight_g") and key[:-2] in sd_in:
            unnormed = True  # weight-norm removed: handled with the matching weight_v entry
        elif key.endswith(".weight_v") and key[:-2] in sd_in:
            t = sd_in[key[:-2]]
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"checkpoint tensor {key[:-2]} has shape {tuple(t.shape)}, expected {shape}")
            out[key[:-2]] = t
            unnormed = True
        elif not is_p:
            out[key] = buffer_value(key, shape)
        else:
            missing.append(key)
    if missing:
        raise KeyError(f"{len(missing)} tensors missing from checkpoint, e.g. {missing[:4]}")
    ema = ckpt.get("ema") if has_wrapper else None
    if ema is not None and spec.ema_decay > 0.0:
        if had_lora or unnormed:
            # the shadow list of such a run follows UniverseLoRA.trainable_parameters (lora.py:135-138), not
            # model_parameters(): it cannot be mapped onto the base model's tensors
            raise NotImplementedError("EMA shadow weights of a LoRA / weight-norm-removed checkpoint are not supported; "
                                      "save the merged model's state_dict without the `ema` entry")
        names = parameter_names(spec)
        shadow = ema["shadow_params"]
        if len(shadow) != len(names):
            raise ValueError(f"EMA has {len(shadow)} shadow params, model has {len(names)} parameters")
        for n, t in zip(names, shadow):
            if tuple(t.shape) != tuple(out[n].shape):
                raise ValueError(f"EMA shadow param for {n}: shape {tuple(t.shape)} != {tuple(out[n].shape)}")
            out[n] = t
    return {k: v.detach().to(torch.float32).cpu().contiguous() for k, v in out.items()}
