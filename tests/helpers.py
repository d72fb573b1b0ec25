"""Shared helpers for the test-suite (model zoo of reduced-width configs, synthetic inputs, unpacking)."""
import json
import math

import torch

import open_universe_amd  # noqa: F401
from open_universe_amd import config as C
from open_universe_amd import state_dict as S

# reduced-width variants of the three shipped topologies (GRU hidden size must stay a multiple of 64)
SMALL = {
    "PP16s": ("PP16", {"score_model.n_channels": 8}),    # OC=128, H=64  (single-workgroup GRU)
    "PP16m": ("PP16", {"score_model.n_channels": 16}),   # OC=256, H=128 (2-workgroup GRU cluster)
    "OR16s": ("OR16", {"score_model.n_channels": 8}),
    "PP24s": ("PP24", {"score_model.n_channels": 8}),
}


def get_spec(name):
    if name in SMALL:
        base, over = SMALL[name]
        return C.spec_from_config(C.builtin_config(base, **over))
    return C.spec_from_config(C.builtin_config(name))


def synth_mix(spec, B, T, seed=1000):
    """SURVEY.md 8(d) synthetic inputs: AM sine + noise."""
    out = []
    t = torch.arange(T) / spec.fs
    for i in range(B):
        g = torch.Generator().manual_seed(seed + i)
        f = 110.0 * (1 + i % 8)
        out.append(0.1 * torch.sin(2 * math.pi * f * t) * (0.5 + 0.5 * torch.sin(2 * math.pi * 3 * t))
                   + 0.03 * torch.randn(T, generator=g))
    return torch.stack(out)


# parameter sets of the signal-transform tests: (tag, stft_kwargs, spec_kwargs, pad_block | None)
TRANSFORM_CASES = [
    ("exp05", dict(n_fft=510, hop_length=128, window_name="hann"),
     dict(transform_type="exponent", abs_exponent=0.5, factor=0.15), None),
    ("log", dict(n_fft=256, hop_length=64, window_name="sqrthann"),
     dict(transform_type="log", abs_exponent=1.0, factor=0.5), None),
    ("none_odd", dict(n_fft=255, hop_length=85, window_name="hamming"),
     dict(transform_type="none", abs_exponent=1.0, factor=1.0), None),
    ("padded", dict(n_fft=510, hop_length=128, window_name="hann"),
     dict(transform_type="exponent", abs_exponent=0.667, factor=0.3), 1024),
]


def varlen_lengths(fs=24000, n=8, seed=5):
    """L_i = fs * U(1, 8) s, manual_seed(5), sorted descending (same recipe as make_golden.py)."""
    u = torch.rand(n, generator=torch.Generator().manual_seed(seed))
    return sorted((int(fs * (1.0 + 7.0 * float(v))) for v in u), reverse=True)


def lora_style_state_dict(sd, rank=4, seed=8):
    """What LoRA fine-tuning of a reference model leaves in a checkpoint (networks/universe/lora.py:96-120,
    lora/utils.py:72-89): keys under `model.`, weight-norm removed (plain `.weight`), Conv1d / Linear adapters already
    merged by `lora.remove()`, ConvTranspose1d adapters still attached (`<p>.conv.weight`, `<p>.lora_weight_a/_b`).
    Returns (that state dict, the equivalent plain state dict with every adapter merged)."""
    g = torch.Generator().manual_seed(seed)
    plain = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            plain[k[:-2]] = torch._weight_norm(v, sd[k[:-2] + "_g"], 0)  # blocks.py:36-42, as remove_weight_norm leaves it
        else:
            plain[k] = v
    merged = dict(plain)
    out = {"model." + k: v for k, v in plain.items()}
    for k, w in plain.items():
        if not (k.endswith("rate_change_conv.conv.weight") and ".decoder.up_modules." in k):
            continue
        if w.shape[0] < rank or w.shape[1] < rank:
            continue
        pfx = k[: -len(".weight")]
        a = 0.05 * torch.randn(w.shape[0], rank, generator=g)
        b = 0.05 * torch.randn(rank, w.shape[1] * w.shape[2], generator=g)
        out["model." + pfx + ".conv.weight"] = out.pop("model." + k)
        if pfx + ".bias" in plain:
            out["model." + pfx + ".conv.bias"] = out.pop("model." + pfx + ".bias")
        out["model." + pfx + ".lora_weight_a"] = a
        out["model." + pfx + ".lora_weight_b"] = b
        merged[k] = w + (a.double() @ b.double()).view(w.shape).float()
    return out, merged


_OBSERVED = {}
_GATES = None


def gate(name, floor=60.0):
    """Pass mark (dB) of a recorded parity figure: tests/parity_gates.json (= what was observed on MI355X minus 15 dB,
    written by tools/make_gates.py), never below `floor` (60 dB = BASELINE.json's tolerance)."""
    global _GATES
    import os

    if _GATES is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parity_gates.json")
        _GATES = json.load(open(path)) if os.path.exists(path) else {}
    return max(float(floor), float(_GATES.get(name, floor)))


def worst(figures):
    """The worst of several parity figures, metric by metric (SI-SDR and plain SNR may bottom out on different rows)."""
    figures = list(figures)
    v = type(figures[0])(min(float(f) for f in figures))
    if hasattr(figures[0], "snr"):
        v.snr = min(f.snr for f in figures)
        v.gain = max(figures, key=lambda f: abs(f.gain - 1.0)).gain
    return v


def record(name, value, floor=60.0):
    """Log an observed parity figure of a GPU test to gpurun_out/parity_observed.json and hold it against its gate: 15 dB
    below what was observed when the gates were last regenerated, so a regression of that size fails instead of hiding
    under the 60 dB bar.  `value` = oracle.restatement.si_sdr(ref, est): SI-SDR in dB, carrying the plain (scale-SENSITIVE)
    SNR of the same pair, which is logged and gated as `<name>#snr` -- a common gain error in the last link (keep_rms
    restore, peak guard, w_out) is invisible to SI-SDR and costs -20 log10 |g - 1| dB of plain SNR."""
    import math
    import os

    checks = [(name, float(value), gate(name, floor))]
    _OBSERVED[name] = round(float(value), 2)
    snr = getattr(value, "snr", None)
    if snr is not None and math.isfinite(snr):
        # (the SNR gate starts from the same floor; comparisons of two HIP variants carry floors of 80-100 dB for both)
        _OBSERVED[name + "#snr"] = round(float(snr), 2)
        checks.append((name + "#snr", float(snr), gate(name + "#snr", floor)))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:  # (logged BEFORE the gate is applied: a figure that fails its gate is the one that has to be on record)
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "parity_observed.json")
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(_OBSERVED)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass
    for n, v, g in checks:
        assert v >= g, f"{n}: {v:.1f} dB < gate {g:.1f} dB (gain of est over ref: {getattr(value, 'gain', float('nan')):.6f})"
    return value


def unpack_conv(blob, L):
    """Recover the dense [M][Cin][KW] weight, bias and PReLU slope of a packed generic-conv layer."""
    Cin, KW, CK, Mp, M = L["Cin"], L["KW"], L["CK"], L["Mp"], L["M"]
    w = blob[L["w_off"]: L["w_off"] + Cin * KW * Mp].view(Cin // CK, KW, CK, Mp)
    W = w.permute(3, 0, 2, 1).reshape(Mp, Cin, KW)[:M]
    bias = blob[L["b_off"]: L["b_off"] + L["Cout"]]
    alpha = blob[L["a_off"]: L["a_off"] + 1]
    return W, bias, alpha


def emulate_conv(blob, L, x, act=True):
    """What conv_mfma_kernel computes for layer L (torch, CPU) -- used to validate packing/folding."""
    import torch.nn.functional as F

    W, bias, alpha = unpack_conv(blob, L)
    if L["act"] and act:
        x = F.prelu(x, alpha)
    if L.get("fir_mode") == 1:  # anti-alias FIR before the strided conv (separate bandwidth pass on the GPU)
        taps = blob[L["fir_off"]: L["fir_off"] + L["fir_len"]]
        x = F.conv1d(x, taps[None, None, :].expand(x.shape[1], 1, -1), padding="same", groups=x.shape[1])
    stride, pad, up, KW = L["stride"], L["pad"], L["up"], L["KW"]
    Nq = x.shape[-1] // stride if stride > 1 else x.shape[-1]
    need = (Nq - 1) * stride + KW
    xp = F.pad(x, (pad, max(0, need - pad - x.shape[-1])))
    y = F.conv1d(xp, W, None, stride=stride)[..., :Nq]
    B = x.shape[0]
    y = y.view(B, L["Cout"], up, Nq).permute(0, 1, 3, 2).reshape(B, L["Cout"], Nq * up)
    y = y + bias.view(1, -1, 1)
    if L.get("fir_mode") == 2:  # FIR + manual bias after the transposed conv
        taps = blob[L["fir_off"]: L["fir_off"] + L["fir_len"]]
        y = F.conv1d(y, taps[None, None, :].expand(y.shape[1], 1, -1), padding="same", groups=y.shape[1])
        y = y + blob[L["fbias_off"]: L["fbias_off"] + L["Cout"]].view(1, -1, 1)
    return y


def plan_convs(plan_json):
    return {c["name"]: c for c in json.loads(plan_json)["convs"]}


def split_copy_as_values(blob, plan_json):
    """The blob with every bf16-split weight copy (conv_split_kernel: bit patterns of three bf16 pieces per weight, NOT fp32
    numbers) replaced by the fp32 values the pieces add up to, piece order kept -- so that two blobs can be compared as numbers
    (a weight that differs in its last bit has completely different mid / lo pieces)."""
    out = blob.clone()
    for L in plan_convs(plan_json).values():
        # (the plain copy, KW taps, and -- round 6 -- the copy of the Winograd-domain weights, KW + 1 transformed taps)
        for on, off, taps in (("ws_on", "ws_off", L["KW"]), ("wsw_on", "wsw_off", L["KW"] + 1)):
            if not L.get(on):
                continue
            n = L["Cin"] * taps * L["Mp"] * 3 // 2
            raw = blob[L[off]: L[off] + n].view(torch.int16)
            pieces = (raw.to(torch.int32) << 16).view(torch.float32).view(-1, 3, 64, 8)   # [fragment triple][piece][lane][8]
            total = pieces.double().sum(dim=1).float()                                     # hi + mid + lo: exact
            filler = torch.zeros(n, dtype=torch.float32)
            filler[: total.numel()] = total.reshape(-1)
            out[L[off]: L[off] + n] = filler
    return out


def experiments_built():
    """True when libouniverse.so was built with `make EXPERIMENTS=1` (conv_block3_kernel, round 1's gru_cluster_kernel and
    the never-selected conv_mfma_kernel configs are in the library only then)."""
    from open_universe_amd import _lib

    return b"+experiments" in _lib.load().ou_version()
