"""
Model specification for the `enhance` hot path.

Mirrors what the reference reads from `config.yaml` -> `config.model`
(reference: config/model/default.yaml, universe_original.yaml, universepp_24k.yaml; consumed at
open_universe/inference_utils/model_loader.py:112-114 and networks/universe/universe.py:45-128).
Only inference-relevant keys are kept; training-only sections (losses, optimizer, scheduler ...) are
ignored.  No hydra / omegaconf: plain PyYAML + `${model.x.y}` interpolation + float coercion
(PyYAML reads `5e-4` as a string).
"""
import math
import re
from dataclasses import asdict, dataclass, field
from typing import List, Optional

import yaml


class DiffKwargs(dict):
    """`model.diff_kwargs` of the reference: supports both attribute and .get() access
    (universe.py:246-249 uses attributes, inference_utils/signature_to_parser.py:47,63 uses .get)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


@dataclass
class NetSpec:
    rate_factors: List[int] = field(default_factory=lambda: [2, 4, 4, 5])
    n_channels: int = 32
    fb_kernel_size: int = 3
    n_rff: int = 32
    noise_cond_dim: int = 512
    extra_conv_block: bool = False
    use_weight_norm: bool = False
    use_antialiasing: bool = False
    time_embedding: Optional[str] = None
    encoder_gru_conv_sandwich: bool = False
    # conditioner only
    n_mels: int = 80
    n_mel_oversample: int = 4
    encoder_gru_residual: bool = False


@dataclass
class ModelSpec:
    kind: str  # "universe" | "universe_gan"
    fs: int
    level_db: float
    edm_noise: Optional[float]
    sigma_min: float
    sigma_max: float
    n_steps: int
    epsilon: float
    schedule: str
    score: NetSpec
    cond: NetSpec
    use_signal_decoupling: bool = False
    signal_decoupling_act: Optional[str] = None
    ema_decay: float = 0.0
    norm_ref: str = "both"                       # normalization_kwargs.ref (only the `target` mode looks at it)
    edm_data_level_db: Optional[float] = None    # edm.data_level_db (universe.py:176-178); None: level_db

    @property
    def tot_ds(self):
        return math.prod(self.score.rate_factors)

    @property
    def score_prefix(self):
        # universe.py:90-95: with EDM the network is registered as `_edm_model`
        return "_edm_model" if self.edm_noise is not None else "score_model"

    def to_dict(self):
        return asdict(self)

    @property
    def diff_kwargs(self):
        return DiffKwargs(schedule=self.schedule, sigma_min=self.sigma_min, sigma_max=self.sigma_max,
                          n_steps=self.n_steps, epsilon=self.epsilon)


_INTERP = re.compile(r"^\$\{([^}]+)\}$")


def _coerce(v):
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            return v
    return v


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.match(node)
        if m:
            cur = root
            for part in m.group(1).split("."):
                if not isinstance(cur, dict) or part not in cur:
                    return None  # training-only interpolation (datamodule / trainer)
                cur = cur[part]
            return _resolve(cur, root)
    return _coerce(node)


def _net_spec(d, is_cond):
    keys = NetSpec.__dataclass_fields__.keys()
    kw = {k: d[k] for k in keys if k in d and d[k] is not None}
    kw["rate_factors"] = [int(r) for r in kw.get("rate_factors", [2, 4, 4, 5])]
    for k in ("n_channels", "fb_kernel_size", "n_rff", "noise_cond_dim", "n_mels", "n_mel_oversample"):
        if k in kw:
            kw[k] = int(kw[k])
    if d.get("seq_model", "gru") != "gru":
        raise NotImplementedError("only seq_model=gru is supported (all shipped configs)")
    if d.get("precoding"):
        raise NotImplementedError("precoding is not used by any shipped config")
    if is_cond and d.get("output_channels") is not None:
        raise NotImplementedError("condition_model.output_channels is None in all shipped configs")
    return NetSpec(**kw)


def spec_from_config(config):
    """`config` = the parsed yaml, either the whole file (with a top-level `model:`) or the model node."""
    root = config if "model" in config else {"model": config}
    m = _resolve(root["model"], root)
    target = str(m.get("_target_", "")).rsplit(".", 1)[-1]
    if target in ("UniverseGAN",):
        kind = "universe_gan"
    elif target in ("Universe",):
        kind = "universe"
    else:
        raise ValueError(f"unsupported model target {m.get('_target_')!r} (Universe | UniverseGAN)")
    if m.get("transform") is not None and kind != "universe_gan":
        raise NotImplementedError("only the identity transform is supported (all shipped configs)")
    nk = m.get("normalization_kwargs", {}) or {}
    if m.get("normalization_norm", 2) not in (2, "2", 2.0) or nk.get("ref", "noisy") not in ("noisy", "both"):
        raise NotImplementedError("only normalization_norm=2 is supported (all shipped configs)")
    edm = m.get("edm")
    diff = m["diffusion"]
    if diff.get("schedule", "geometric") != "geometric":
        raise NotImplementedError(f"schedule {diff.get('schedule')}")  # universe.py:380-386
    losses = m.get("losses", {}) or {}
    score = _net_spec(m["score_model"], False)
    cond = _net_spec(m["condition_model"], True)
    if cond.rate_factors != score.rate_factors or cond.n_channels != score.n_channels:
        raise ValueError("score and condition networks must share rate_factors / n_channels")
    return ModelSpec(
        kind=kind,
        fs=int(m["fs"]),
        level_db=float(nk.get("level_db", 0.0)),
        edm_noise=float(edm["noise"]) if edm is not None else None,
        sigma_min=float(diff["sigma_min"]),
        sigma_max=float(diff["sigma_max"]),
        n_steps=int(diff["n_steps"]),
        epsilon=float(diff["epsilon"]),
        schedule="geometric",
        score=score,
        cond=cond,
        use_signal_decoupling=bool(kind == "universe_gan" and losses.get("use_signal_decoupling", False)),
        signal_decoupling_act=losses.get("signal_decoupling_act") if kind == "universe_gan" else None,
        ema_decay=float((m.get("training") or {}).get("ema_decay", 0.0) or 0.0),
        norm_ref=str(nk.get("ref", "noisy")),  # utils/norm.py:47 default
        edm_data_level_db=(float(edm["data_level_db"]) if edm is not None and edm.get("data_level_db") is not None
                           else None),
    )


def load_config(path):
    with open(path, "r") as f:
        return yaml.safe_load(f)


# the three shipped model configurations (PP16 / OR16 / PP24), for synthetic-weight use
def builtin_config(name, **over):
    base = dict(
        fs=16000, normalization_norm=2, normalization_kwargs=dict(ref="both", level_db=-26.0),
        diffusion=dict(schedule="geometric", sigma_min=5e-4, sigma_max=5.0, n_steps=8, epsilon=1.3),
        training=dict(ema_decay=0.999),
    )
    net = dict(fb_kernel_size=3, rate_factors=[2, 4, 4, 5], n_channels=32, n_rff=32, noise_cond_dim=512,
               encoder_gru_conv_sandwich=False, extra_conv_block=True)
    cnet = dict(n_mels=80, n_mel_oversample=4, encoder_gru_residual=True, use_antialiasing=False,
                time_embedding=None)
    if name in ("PP16", "default", "universepp_16k"):
        s = dict(net, use_weight_norm=True, use_antialiasing=True, time_embedding="simple")
        model = dict(base, _target_="open_universe.networks.universe.UniverseGAN", edm=dict(noise=0.25),
                     score_model=s, condition_model=dict(s, **cnet),
                     losses=dict(use_signal_decoupling=True, signal_decoupling_act="snake"))
    elif name in ("OR16", "universe_original"):
        s = dict(net, use_weight_norm=False, use_antialiasing=False)
        model = dict(base, _target_="open_universe.networks.universe.Universe",
                     score_model=s, condition_model=dict(s, **cnet))
    elif name in ("PP24", "universepp_24k"):
        s = dict(net, rate_factors=[2, 3, 5, 8], n_channels=48, use_weight_norm=True,
                 use_antialiasing=True, time_embedding="simple")
        model = dict(base, fs=24000, _target_="open_universe.networks.universe.UniverseGAN",
                     edm=dict(noise=0.25), score_model=s,
                     condition_model=dict(s, **dict(cnet, n_mels=128)),
                     losses=dict(use_signal_decoupling=True, signal_decoupling_act="snake"))
    else:
        raise ValueError(name)
    for key, val in over.items():
        node = model
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = val
        if parts[0] == "score_model" and parts[-1] in ("rate_factors", "n_channels", "extra_conv_block",
                                                          "use_weight_norm", "fb_kernel_size"):
            model["condition_model"][parts[-1]] = val
    return {"model": model}
