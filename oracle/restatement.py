"""
TEST INFRASTRUCTURE -- the ORACLE.  Not part of the product path.

CPU restatement (plain PyTorch fp32 functional ops) of the reference's `model.enhance` hot path:
conditioner network, score network, EDM wrapper and the reverse-diffusion sampler of UNIVERSE /
UNIVERSE++ (line/open-universe).  Every function cites the reference file:line it follows (paths
relative to the reference checkout).  It consumes a state-dict with the reference's own key schema and a
plain `spec` dict (see `open_universe_amd.config.ModelSpec.to_dict`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module, and only
as the checker / the timed CPU baseline -- never as the thing shipped.  The product path
(`open_universe_amd`) does not import it and fails loudly when the HIP library is missing.

Pinning: validated in the build container against the *imported* reference itself
(`oracle/ref_import.py`, `tests/test_oracle_vs_reference.py`) and against the committed golden vectors
generated from that import (`tests/golden/`).  The reference has no tests / golden vectors of its own
(SURVEY.md section 4).  torchaudio (MelSpectrogram, Resample) is absent from the image: those two pieces are
restated from torchaudio's documented algorithm and are therefore "parity unpinned" at that third-party
boundary (the mel window / filterbank / resample kernels are checkpoint buffers, so with a real checkpoint
their values come from the file).
"""
import math

import torch
import torch.nn.functional as F

INV_SQRT2 = 1.0 / math.sqrt(2)


# --------------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------------
def eff_weight(sd, p):
    """Effective weight of a (possibly weight-normed) Conv1d/ConvTranspose1d/Linear at prefix `p`.

    blocks.py:36-42 cond_weight_norm -> old-style torch.nn.utils.weight_norm(dim=0):
    w = v * (g / ||v||), norm over all dims but 0.  The reference recomputes this every forward."""
    if p + ".weight_g" in sd:
        g, v = sd[p + ".weight_g"], sd[p + ".weight_v"]
        return torch._weight_norm(v, g, 0)
    return sd[p + ".weight"]


def binomial_filter(kernel_size):
    """blocks.py:62-68 get_binomial_filter: Pascal row, scaled to unit RMS (sum w^2 = K)."""
    row = [float(math.comb(kernel_size - 1, i)) for i in range(kernel_size)]
    w = torch.tensor(row, dtype=torch.float64)
    full = torch.zeros(kernel_size, kernel_size, dtype=torch.float64)
    for n in range(kernel_size):
        for i in range(n + 1):
            full[n, i] = math.comb(n, i)
    norm = full.square().mean().sqrt()
    w = (w / norm).to(torch.float32)
    w = w / w.square().mean().sqrt()
    return w


# --------------------------------------------------------------------------------------------------
# blocks.py
# --------------------------------------------------------------------------------------------------
def film(x, y):
    """blocks.py:53-59."""
    c = x.shape[1]
    y = y.view(y.shape + (1,) * (x.ndim - y.ndim))
    return y[:, :c] * x + y[:, c:]


def anti_alias(sd, p, x):
    """blocks.py:119-130 BinomialAntiAlias: depthwise 'same' conv with the buffered taps."""
    w = sd[p + ".weights"]
    c = x.shape[1]
    return F.conv1d(x, w[None, None, :].expand(c, 1, -1), padding="same", groups=c)


def prelu_conv(sd, p, x, stride=1, transpose=False, same=False, act="prelu"):
    """blocks.py:205-227 PReLU_Conv.forward."""
    r = x.shape[-1] % stride
    if not transpose and r != 0:
        x = F.pad(x, (0, stride - r))
    if act == "prelu":
        x = F.prelu(x, sd[p + ".prelu.weight"])
    elif act == "snake":
        x = alias_free_snake(sd, p + ".prelu", x)
    aa = (p + ".low_pass_filter.weights") in sd
    if aa and not transpose:
        x = anti_alias(sd, p + ".low_pass_filter", x)
    w = eff_weight(sd, p + ".conv")
    b = sd.get(p + ".conv.bias", None)
    if transpose:
        x = F.conv_transpose1d(x, w, b, stride=stride)
    else:
        x = F.conv1d(x, w, b, stride=stride, padding="same" if same else 0)
    if aa and transpose:
        x = anti_alias(sd, p + ".low_pass_filter", x)
    if (p + ".bias") in sd:
        x = x + sd[p + ".bias"].reshape(1, -1, 1)
    return x


def conv_block(sd, p, h, rate=None, direction="none", noise_cond=None, input_cond=None, res=None,
               length=None):
    """blocks.py:327-412 ConvBlock.forward.  Returns (h_next, v_out, cond_out)."""
    if direction == "up":
        if length is not None and rate * h.shape[-1] < length:
            h = F.pad(h, (0, 1))
        h = prelu_conv(sd, p + ".rate_change_conv", h, stride=rate, transpose=True)
        if length is not None:
            h = F.pad(h, (0, length - h.shape[-1]))
    if res is not None:
        h = (h + res) * INV_SQRT2
    cond_out = prelu_conv(sd, p + ".conv1", h, same=True)
    c = cond_out
    if input_cond is not None:
        c = (cond_out + input_cond) * INV_SQRT2
    if noise_cond is not None:
        c = film(c, noise_cond)
    c = prelu_conv(sd, p + ".conv2", c, same=True)
    c = prelu_conv(sd, p + ".conv3", c, same=True)
    v_out = (h + c) * INV_SQRT2
    if direction == "down":
        r = h.shape[-1] % rate
        v_pad = F.pad(v_out, (0, rate - r)) if r != 0 else v_out
        return prelu_conv(sd, p + ".rate_change_conv", v_pad, stride=rate), v_out, cond_out
    return v_out, v_out, cond_out


# --------------------------------------------------------------------------------------------------
# GRU (torch.nn.GRU, batch_first, bidirectional) -- score.py:83-89,116; condition.py:173-179,212
# --------------------------------------------------------------------------------------------------
def gru_cell_sequence(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """Explicit recurrence of one GRU direction (PyTorch gate order r, z, n; h0 = 0).
    x: (B, T, I) -> (B, T, H)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = x @ w_ih.t() + b_ih
    h = x.new_zeros(B, H)
    out = x.new_zeros(B, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gx[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gx[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gx[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (h - n) * z + n
        out[:, t] = h
    return out


def gru(sd, p, x, num_layers=1, explicit=False):
    """x: (B, C, T) channel-major (the transposes of score.py:116-117 are folded in)."""
    y = x.transpose(-2, -1)
    if explicit:
        for layer in range(num_layers):
            outs = []
            for sfx, rev in (("", False), ("_reverse", True)):
                k = f"_l{layer}{sfx}"
                outs.append(gru_cell_sequence(y, sd[p + ".weight_ih" + k], sd[p + ".weight_hh" + k],
                                              sd[p + ".bias_ih" + k], sd[p + ".bias_hh" + k], rev))
            y = torch.cat(outs, dim=-1)
    else:
        flat = []
        for layer in range(num_layers):
            for sfx in ("", "_reverse"):
                k = f"_l{layer}{sfx}"
                flat += [sd[p + ".weight_ih" + k], sd[p + ".weight_hh" + k],
                         sd[p + ".bias_ih" + k], sd[p + ".bias_hh" + k]]
        H = flat[1].shape[1]
        h0 = y.new_zeros(2 * num_layers, y.shape[0], H)
        y, _ = torch._VF.gru(y.contiguous(), h0, flat, True, num_layers, 0.0, False, True, True)
    return y.transpose(-2, -1)


# --------------------------------------------------------------------------------------------------
# sigma_block.py
# --------------------------------------------------------------------------------------------------
def simple_time_embedding(sd, p, log10_sigma, n_dim):
    """sigma_block.py:73-78."""
    time = torch.arange(n_dim // 2)
    f = 0.5 * torch.sigmoid(sd[p + ".weight"] * log10_sigma[:, None] + sd[p + ".bias"])
    ph = 2.0 * math.pi * f * time
    return torch.cat([torch.sin(ph), torch.cos(ph)], dim=-1)


def sigma_block_rff(sd, p, log10_sigma):
    """sigma_block.py:50-57 (+ Linear_PReLU :24-33)."""
    ph = 2.0 * math.pi * sd[p + ".freq"][None, :] * log10_sigma[:, None]
    g = torch.cat([torch.sin(ph), torch.cos(ph)], dim=-1)
    for i in (1, 2, 3):
        q = f"{p}.layer{i}"
        g = F.prelu(F.linear(g, sd[q + ".lin.weight"], sd[q + ".lin.bias"]), sd[q + ".prelu.weight"])
    return g


# --------------------------------------------------------------------------------------------------
# score.py
# --------------------------------------------------------------------------------------------------
def score_network(sd, p, spec, x, sigma, cond, taps=None):
    """score.py:277-297 ScoreNetwork.forward (encoder :104-127, decoder :196-210)."""
    s = spec["score"]
    rates = list(s["rate_factors"])
    n_samples = x.shape[-1]
    log_s = torch.log10(sigma)
    if s.get("time_embedding") == "simple":
        g = simple_time_embedding(sd, p + ".sigma_block", log_s, s["noise_cond_dim"])
    else:
        g = sigma_block_rff(sd, p + ".sigma_block", log_s)
    x = F.conv1d(x, sd[p + ".input_conv.weight"], sd[p + ".input_conv.bias"], padding="same")
    if taps is not None:
        taps["g"] = g
        taps["input_conv"] = x
    residuals, lengths = [], []
    n_blocks = len(rates) + (1 if s["extra_conv_block"] else 0)
    for i in range(n_blocks):
        q = f"{p}.encoder"
        nc = F.linear(g, eff_weight(sd, f"{q}.cond_proj.{i}"), sd[f"{q}.cond_proj.{i}.bias"])
        lengths.append(x.shape[-1])
        if i < len(rates):
            x, res, _ = conv_block(sd, f"{q}.ds_modules.{i}", x, rates[i], "down", noise_cond=nc)
        else:
            x, res, _ = conv_block(sd, f"{q}.ds_modules.{i}", x, noise_cond=nc)
        residuals.append(res)
        if taps is not None:
            taps[f"enc{i}.v"] = res
            taps[f"enc{i}.h"] = x
    x = gru(sd, p + ".encoder.gru", x, 1)
    if taps is not None:
        taps["gru"] = x
    residuals, lengths = residuals[::-1], lengths[::-1]
    up = rates[::-1]
    for j in range(n_blocks):
        q = f"{p}.decoder"
        nc = F.linear(g, eff_weight(sd, f"{q}.noise_cond_proj.{j}"), sd[f"{q}.noise_cond_proj.{j}.bias"])
        sc = F.conv1d(cond[j], eff_weight(sd, f"{q}.signal_cond_proj.{j}"), sd[f"{q}.signal_cond_proj.{j}.bias"])
        if s["extra_conv_block"]:
            rate, direction = (None, "none") if j == 0 else (up[j - 1], "up")
        else:
            rate, direction = up[j], "up"
        x, _, _ = conv_block(sd, f"{q}.up_modules.{j}", x, rate, direction, noise_cond=nc,
                             input_cond=sc, res=residuals[j], length=lengths[j])
        if taps is not None:
            taps[f"dec{j}.v"] = x
    x = F.prelu(x, sd[p + ".prelu.weight"])
    x = prelu_conv(sd, p + ".output_conv", x, same=True)
    return F.pad(x, (0, n_samples - x.shape[-1]))


# --------------------------------------------------------------------------------------------------
# condition.py
# --------------------------------------------------------------------------------------------------
def mel_spec(sd, p, spec, x):
    """condition.py:92-108 MelAdapter.compute_mel_spec (+ torchaudio MelSpectrogram, restated:
    periodic Hann, center=False, power 2, HTK filterbank taken from the state-dict buffers)."""
    c = spec["cond"]
    hop = math.prod(spec["score"]["rate_factors"])
    n_fft = c["n_mel_oversample"] * hop
    pad_tot = n_fft - hop
    pad_left, pad_right = pad_tot // 2, pad_tot - pad_tot // 2
    r = x.shape[-1] % hop
    pad = hop - r if r != 0 else 0
    x = F.pad(x, (pad_left, pad + pad_right))
    shape = x.shape
    st = torch.stft(x.reshape(-1, shape[-1]), n_fft, hop, n_fft, sd[p + ".mel_spec.spectrogram.window"],
                    center=False, onesided=True, normalized=False, return_complex=True)
    pw = st.abs().pow(2.0)
    mel = torch.matmul(pw.transpose(-1, -2), sd[p + ".mel_spec.mel_scale.fb"]).transpose(-1, -2)
    mel = mel.reshape(shape[0], -1, mel.shape[-1])  # squeeze(1) of the single channel
    norm = (mel ** 2).sum(dim=-2, keepdim=True).mean(dim=-1, keepdim=True).sqrt()
    return mel / norm.clamp(min=1e-5)


def conditioner_network(sd, p, spec, x, x_wav=None, taps=None):
    """condition.py:346-377 ConditionerNetwork.forward(train=True) -> (conditions, y_hat, h)."""
    c = spec["cond"]
    rates = list(spec["score"]["rate_factors"])
    n_samples = x.shape[-1]
    if x_wav is None:
        x_wav = x
    # MelAdapter.forward  condition.py:110-114
    m = mel_spec(sd, p + ".input_mel", spec, x_wav)
    if taps is not None:
        taps["mel"] = m
    m = F.conv1d(m, eff_weight(sd, p + ".input_mel.conv"), sd[p + ".input_mel.conv.bias"], padding="same")
    x_mel, _, _ = conv_block(sd, p + ".input_mel.conv_block", m)
    if taps is not None:
        taps["x_mel"] = x_mel
    x = F.conv1d(x, eff_weight(sd, p + ".input_conv"), sd[p + ".input_conv.bias"], padding="same")
    # ConditionerEncoder.forward  condition.py:189-220
    outputs, lengths = [], []
    n_blocks = len(rates) + (1 if c["extra_conv_block"] else 0)
    st_rates = [math.prod(rates[i:]) for i in range(len(rates))]
    for i in range(n_blocks):
        lengths.append(x.shape[-1])
        q = f"{p}.encoder.ds_modules.{i}"
        if i < len(rates):
            x, res, _ = conv_block(sd, q, x, rates[i], "down")
        else:
            x, res, _ = conv_block(sd, q, x)
        if i < len(rates) - 1:
            o = prelu_conv(sd, f"{p}.encoder.st_convs.{i}", res, stride=st_rates[i])
            outputs.append(o)
            if taps is not None:
                taps[f"st{i}"] = o
    outputs.append(x)
    out = x_mel
    for o in outputs:
        out = out + o
    out = out * (1.0 / math.sqrt(len(outputs) + 1))
    if taps is not None:
        taps["enc_sum"] = out
    out, _, _ = conv_block(sd, p + ".encoder.conv_block1", out)
    res = out
    out = gru(sd, p + ".encoder.gru", out, 2)
    if c["encoder_gru_residual"]:
        out = (out + res) / math.sqrt(2)
    if taps is not None:
        taps["gru"] = out
    h, _, _ = conv_block(sd, p + ".encoder.conv_block2", out)
    lengths = lengths[::-1]
    # ConditionerDecoder.forward  condition.py:264-270
    conditions = []
    y, _, _ = conv_block(sd, p + ".decoder.input_conv_block", h)
    up = rates[::-1]
    for j in range(n_blocks):
        if c["extra_conv_block"]:
            rate, direction = (None, "none") if j == 0 else (up[j - 1], "up")
        else:
            rate, direction = up[j], "up"
        y, _, cond = conv_block(sd, f"{p}.decoder.up_modules.{j}", y, rate, direction, length=lengths[j])
        conditions.append(cond)
    y = F.pad(y, (0, n_samples - y.shape[-1]))
    return conditions, y, h


# --------------------------------------------------------------------------------------------------
# bigvgan/snake.py + alias_free_act.py (cold branch: use_aux_signal / warm_start)
# --------------------------------------------------------------------------------------------------
def _apply_resample(x, kernel, orig, new):
    """torchaudio.functional._apply_sinc_resample_kernel (restated)."""
    width = (kernel.shape[-1] - orig) // 2
    shape = x.shape
    x = x.reshape(-1, shape[-1])
    length = x.shape[-1]
    x = F.pad(x, (width, width + orig))
    r = F.conv1d(x[:, None], kernel, stride=orig)
    r = r.transpose(1, 2).reshape(x.shape[0], -1)
    r = r[..., : int(math.ceil(new * length / orig))]
    return r.reshape(shape[:-1] + r.shape[-1:])


def alias_free_snake(sd, p, x):
    """alias_free_act.py:8-30 Activation1d(2x up -> Snake(log-scale alpha) snake.py:53-64 -> 2x down)."""
    x = _apply_resample(x, sd[p + ".act.upsample.kernel"], 1, 2)
    alpha = torch.exp(sd[p + ".act.act.alpha"])[None, :, None]
    x = x + (1.0 / (alpha + 1e-9)) * torch.pow(torch.sin(x * alpha), 2)
    return _apply_resample(x, sd[p + ".act.downsample.kernel"], 2, 1)


def aux_to_wav(sd, spec, y_aux):
    """universe_gan.py:145-149 (UniverseGAN) / universe.py:228-229 (Universe)."""
    if spec.get("use_signal_decoupling"):
        return prelu_conv(sd, "signal_decoupling_layer", y_aux, same=True,
                          act=spec.get("signal_decoupling_act") or "none")
    return y_aux


# --------------------------------------------------------------------------------------------------
# utils/norm.py, universe.py
# --------------------------------------------------------------------------------------------------
def normalize(x, level_db, eps=1e-5):
    """utils/norm.py:47-87 normalize_batch(norm=2, ref='both', zero_mean=True) for one tensor."""
    level = 10 ** (level_db / 20.0)
    x = x - x.mean(dim=(1, 2), keepdim=True)
    gain = level / x.std(dim=(1, 2), keepdim=True).clamp(min=eps)
    return x * gain


def edm_weights(spec, sigma):
    """universe.py:175-189 (sigma_data from edm.data_level_db when given, else normalization_kwargs.level_db)."""
    level_db = spec.get("edm_data_level_db")
    sigma_data = 10.0 ** ((spec["level_db"] if level_db is None else level_db) / 20.0)
    sigma_norm = (sigma ** 2 + sigma_data ** 2) ** 0.5
    return {
        "skip": sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2),
        "in": 1.0 / sigma_norm,
        "out": sigma * sigma_data / sigma_norm,
        "noise": spec["edm_noise"],
    }


def score_model(sd, spec, x, sigma, cond):
    """universe.py:197-209 (_edm_score_wrapper) for UNIVERSE++; plain score net otherwise."""
    if spec.get("edm_noise") is not None:
        w = edm_weights(spec, sigma)
        b3 = lambda a: a[:, None, None]
        net = score_network(sd, "_edm_model", spec, b3(w["in"]) * x, w["noise"] * sigma, cond)
        est = b3(w["skip"]) * x + b3(w["out"]) * net
        return (est - x) / b3(sigma) ** 2
    return score_network(sd, "score_model", spec, x, sigma, cond)


def sampler_constants(spec, n_steps, epsilon):
    """universe.py:301-311: (sigma[n] fp32 tensor, eta, beta)."""
    delta_t = 1.0 / (n_steps - 1)
    gamma = (spec["sigma_max"] / spec["sigma_min"]) ** -delta_t
    eta = 1 - gamma ** epsilon
    beta = math.sqrt(1 - gamma ** (2 * (epsilon - 1.0)))
    time = torch.linspace(0, 1, n_steps).to(torch.float32).flip(dims=[0])
    sigma = spec["sigma_min"] * (spec["sigma_max"] / spec["sigma_min"]) ** time
    return sigma, eta, beta


def randn(x, sigma, rng=None):
    """universe.py:39-41."""
    return torch.randn(x.shape, dtype=x.dtype, device=x.device, generator=rng) * sigma[:, None, None]


def signal_median(signal):
    """utils/stats.py:22-66."""
    shape = signal.shape
    signal = signal.flatten(start_dim=2)
    n = signal.shape[0]
    _, sorted_indices = signal.sort(dim=0)
    _, min_indices = abs(sorted_indices - n / 2).min(dim=0)
    pad_bins = torch.arange(n)[None, :].expand(min_indices.shape[0], n)
    min_indices = torch.cat((min_indices, pad_bins), dim=1)
    counts = torch.cat([(min_indices == i).sum(dim=1, keepdim=True) for i in range(n)], dim=1) - 1
    select = counts.argmax(dim=1)
    med = torch.stack([signal[select[i], i, :] for i in range(signal.shape[1])], dim=0)
    return med.reshape(shape[1:])


@torch.no_grad()
def enhance(sd, spec, mix, n_steps=None, epsilon=None, target=None, fake_score_snr=None, rng=None,
            use_aux_signal=False, keep_rms=False, ensemble=None, ensemble_stat="median", warm_start=None,
            noise=None):
    """universe.py:231-375 Universe.enhance.

    `noise`: optional list of pre-drawn standard-normal tensors (B,1,T_pad) consumed instead of `rng`
    draws, in the reference's draw order (x0, z_0 ... z_{N-2}); lets the HIP path and the oracle share
    the exact same noise regardless of device."""
    if epsilon is None:
        epsilon = spec["epsilon"]
    if n_steps is None:
        n_steps = spec["n_steps"]
    x_ndim = mix.ndim
    if x_ndim == 1:
        mix = mix[None, None, :]
    elif x_ndim == 2:
        mix = mix[:, None, :]
    elif x_ndim > 3:
        raise ValueError("The input should have at most 3 dimensions")
    mix_rms = mix.square().mean(dim=(-2, -1), keepdim=True).sqrt()
    if ensemble is not None:
        mix_shape = mix.shape
        mix = torch.stack([mix] * ensemble, dim=0).view((-1,) + mix_shape[1:])
    tot_ds = math.prod(spec["score"]["rate_factors"])
    mix_len = mix.shape[-1]
    pad = tot_ds - mix_len % tot_ds  # universe.py:219-223 (a full block when already a multiple)
    mix = F.pad(mix, (pad // 2, pad - pad // 2))
    if target is not None:
        target = F.pad(target, (pad // 2, pad - pad // 2))
        if spec.get("norm_ref", "both") == "both":  # utils/norm.py:77-82: the target by its own statistics
            target = normalize(target, spec["level_db"])
        else:  # utils/norm.py:83-84 ref == "noisy": (tgt - mean_mix) * gain_mix
            mean = mix.mean(dim=(1, 2), keepdim=True)
            gain = 10 ** (spec["level_db"] / 20.0) / (mix - mean).std(dim=(1, 2), keepdim=True).clamp(min=1e-5)
            target = (target - mean) * gain
    mix = normalize(mix, spec["level_db"])
    score_snr = 5.0 if fake_score_snr is None else fake_score_snr
    noise_iter = iter(noise) if noise is not None else None

    def draw(ref, sig):
        if noise_iter is not None:
            return next(noise_iter) * sig[:, None, None]
        return randn(ref, sig, rng)

    def score_wrapper(x, s, cond):
        if target is None:
            return score_model(sd, spec, x, s, cond)
        true_score = -(x - target) / s[:, None, None] ** 2
        noise_rms = (true_score ** 2).mean().sqrt() * 10 ** (-score_snr / 20.0)
        nz = torch.randn(true_score.shape, dtype=true_score.dtype, generator=rng)
        return true_score + nz * noise_rms

    sigma, eta, beta = sampler_constants(spec, n_steps, epsilon)
    sigma = sigma[None, :].expand(mix.shape[0], -1)
    cond, aux_signal, _ = conditioner_network(sd, "condition_model", spec, mix, x_wav=mix)
    if use_aux_signal:
        x = aux_to_wav(sd, spec, aux_signal)
    else:
        if warm_start is None:
            x = draw(mix, sigma[:, 0])
            n_start = 0
        else:
            sig = aux_to_wav(sd, spec, aux_signal)
            x = sig + draw(sig, sigma[:, warm_start])
            n_start = warm_start
        for n in range(n_start, n_steps - 1):
            s_now, s_next = sigma[:, n], sigma[:, n + 1]
            score = score_wrapper(x, s_now, cond)
            z = draw(x, s_next)
            x = x + s_now[..., None, None] ** 2 * eta * score + beta * z
        score = score_wrapper(x, sigma[:, -1], cond)
        x = x + sigma[:, -1, None, None] ** 2 * score
    x = x[..., pad // 2: -(pad - pad // 2)]
    x = F.pad(x, (0, mix_len - x.shape[-1]))
    if keep_rms:
        x_rms = x.square().mean(dim=(-2, -1), keepdim=True).sqrt().clamp(min=1e-5)
        x = x * (mix_rms / x_rms)
    scale = abs(x).max(dim=-1, keepdim=True).values
    x = torch.where(scale > 1.0, x / scale, x)
    if ensemble is not None:
        x = x.view((-1,) + mix_shape)
        if ensemble_stat == "mean":
            x = x.mean(dim=0)
        elif ensemble_stat == "median":
            x = x.median(dim=0).values
        elif ensemble_stat == "signal_median":
            x = signal_median(x)
        else:
            raise NotImplementedError()
    if x_ndim == 1:
        x = x[0, 0]
    elif x_ndim == 2:
        x = x[:, 0, :]
    return x


# --------------------------------------------------------------------------------------------------
# layers/dyn_range_comp.py (signal transforms; no shipped config uses them)
# --------------------------------------------------------------------------------------------------
def compressed_mag_stft(x, n_fft, hop_length, window, transform_type, abs_exponent, factor, inv=False, length=None,
                        pad_block=None):
    """dyn_range_comp.py:51-225 CompressedMagSTFT / CompressedMagSTFTPadded (pad_block is not None -> Padded, whose
    `_stft` applies `_pad` twice, :199-201).  window: the n_fft-tap analysis / synthesis window."""
    padded = pad_block is not None

    def _pad(sig):  # :182-196
        if pad_block:
            r = sig.shape[-1] % pad_block
            if r > 0:
                sig = F.pad(sig, (0, pad_block - r))
        return sig[..., :-hop_length]

    if not inv:
        sig = x.squeeze(1)
        if padded:
            sig = _pad(_pad(sig))
        spec = torch.stft(sig, n_fft=n_fft, hop_length=hop_length, window=window, center=True, return_complex=True,
                          pad_mode="constant")
        if transform_type == "exponent":  # :117-124
            if abs_exponent != 1:
                spec = (1e-7 + spec.abs()) ** (abs_exponent - 1.0) * spec
            spec = spec * factor
        elif transform_type == "log":  # :125-127
            spec = torch.log(1 + spec.abs()) * torch.sgn(spec) * factor
        out = torch.view_as_real(spec).moveaxis(3, 1)  # (batch, real/imag, freq, time)  :91-95
        return out.flatten(start_dim=1, end_dim=2)
    n_freq = x.shape[1] // 2
    spec = torch.view_as_complex(x.reshape(x.shape[0], 2, n_freq, x.shape[2]).moveaxis(1, 3).contiguous())  # :104-106
    if transform_type == "exponent":  # :133-139
        spec = spec / factor
        if abs_exponent != 1:
            spec = (1e-7 + spec.abs()) ** (1.0 / abs_exponent - 1.0) * spec
    elif transform_type == "log":  # :140-142
        spec = spec / factor
        spec = (torch.exp(spec.abs()) - 1) * torch.sgn(spec)
    if padded and length is None:
        length = spec.shape[-1] * hop_length  # :215-216
    y = torch.istft(spec, n_fft=n_fft, hop_length=hop_length, window=window, center=True, length=length)
    return y.unsqueeze(1)


class ParityFigure(float):
    """SI-SDR in dB (the float itself) that also carries the figures a scale-invariant measure cannot see:
    `.snr` = plain SNR 10 log10(|ref|^2 / |est - ref|^2) (a common gain error g shows up as -20 log10 |g - 1|) and
    `.gain` = the least-squares gain of `est` over `ref` that SI-SDR projects away."""
    snr = float("nan")
    gain = float("nan")


def snr_db(ref, est):
    """Plain (NOT scale-invariant) SNR in dB of `est` against `ref`."""
    ref = ref.reshape(-1).double()
    est = est.reshape(-1).double()
    return float(10 * torch.log10(ref.square().sum().clamp(min=1e-300) / (est - ref).square().sum().clamp(min=1e-300)))


def si_sdr(ref, est):
    """Scale-invariant SDR in dB of `est` against `ref` (the parity gate: >= 60 dB).  The returned float carries the plain
    SNR and the projected gain as attributes (ParityFigure); tests/helpers.record() logs and gates both."""
    ref = ref.reshape(-1).double()
    est = est.reshape(-1).double()
    a = (ref @ est) / (ref @ ref).clamp(min=1e-30)
    err = est - a * ref
    v = ParityFigure(10 * torch.log10((a * ref).square().sum() / err.square().sum().clamp(min=1e-300)))
    v.snr = snr_db(ref, est)
    v.gain = float(a)
    return v
