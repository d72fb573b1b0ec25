"""
`load_model` with the reference's signature and error behaviour
(open_universe/inference_utils/model_loader.py:62-137), returning the MI355X-native model object.

Differences, all forced by the target: the model lives on a HIP device (device=None -> current GPU; a CPU
device raises), `instantiate(config.model)` is replaced by `spec_from_config` + the C-ABI packer, and
weight-norm / EMA are resolved once at load time instead of on every forward.

The checkpoint is a pickle.  The reference calls `torch.load(ckpt_path, map_location=device)` on whatever
`hf_hub_download` returned; here nothing but tensors is ever materialised: first `weights_only=True`, and for Lightning
files whose `hyper_parameters` carry omegaconf / lightning objects (which that mode rejects) a restricted unpickler
that resolves only the tensor-rebuilding globals and replaces every other global by an inert placeholder -- no
constructor or `__reduce__` target from the file is ever executed.  Full unpickling needs OU_UNSAFE_PICKLE=1.
"""
import collections
import os
import pickle
from pathlib import Path

import torch

from ..config import load_config, spec_from_config
from ..state_dict import inference_state_dict, merge_lora, model_schema
from ..universe import Universe, UniverseGAN

supported_models = ["universe"]

# prefixes of modules that only exist for training (universe.py:138-172, universe_gan.py:95-116)
TRAINING_ONLY_PREFIXES = ("loss_", "enh_losses.", "losses.")


def ckpt_to_config_path(ckpt_path):
    """model_loader.py:33-48: <ckpt dir>/config.yaml or <ckpt dir>/../.hydra/config.yaml."""
    ckpt_path = Path(ckpt_path)
    candidates = [ckpt_path.parent / "config.yaml", ckpt_path.parents[1] / ".hydra/config.yaml"]
    for c in candidates:
        if c.exists():
            return c
    raise ValueError(f"Could not find the configuration file for model {ckpt_path}.")


class _Inert:
    """Placeholder for any global the restricted unpickler refuses to import: absorbs construction and state."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __setstate__(self, state):
        pass

    def __setitem__(self, k, v):
        pass

    def append(self, v):
        pass

    def extend(self, v):
        pass

    def update(self, *a, **k):
        pass


_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("torch._utils", "_rebuild_tensor_v2"): torch._utils._rebuild_tensor_v2,
    ("torch._utils", "_rebuild_parameter"): torch._utils._rebuild_parameter,
    ("torch", "Size"): torch.Size,
    ("torch", "device"): torch.device,
}
for _n in ("FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage",
           "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage"):
    _ALLOWED_GLOBALS[("torch", _n)] = getattr(torch, _n)
for _n in ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool"):
    _ALLOWED_GLOBALS[("torch", _n)] = getattr(torch, _n)


class _TensorOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        return _ALLOWED_GLOBALS.get((module, name), _Inert)


class _tensor_only_pickle:
    """`pickle_module` for torch.load: tensors and containers only."""
    __name__ = "ou_tensor_only_pickle"
    Unpickler = _TensorOnlyUnpickler
    load = staticmethod(lambda f, **kw: _TensorOnlyUnpickler(f, **kw).load())


def read_checkpoint(path):
    """-> the checkpoint dict with every tensor on the CPU; no code from the file is executed."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        pass
    if os.environ.get("OU_UNSAFE_PICKLE") == "1":
        return torch.load(path, map_location="cpu", weights_only=False)
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_tensor_only_pickle)


def _resolve(ckpt_path, hf_token):
    """-> (checkpoint path, config path): a local file with config.yaml beside it, or a Huggingface id repo[:revision]."""
    if not Path(ckpt_path).exists():
        try:
            from huggingface_hub import hf_hub_download

            ckpt_path = str(ckpt_path)
            repo_id, _, revision = ckpt_path.partition(":")
            kw = dict(repo_id=repo_id, revision=revision or None, token=hf_token)
            ckpt_path = hf_hub_download(filename="weights.ckpt", **kw)
            config_path = hf_hub_download(filename="config.yaml", **kw)
        except Exception as e:
            print(f"{ckpt_path} is not a local file and download from HF hub failed.")
            raise e
    else:
        ckpt_path = Path(ckpt_path)
        config_path = ckpt_to_config_path(ckpt_path)
    return ckpt_path, config_path


def _inference_weights(spec, ckpt_path, strict):
    """The tensors `enhance` runs on (EMA weights when the checkpoint has them), after the reference's strictness rule."""
    data = read_checkpoint(ckpt_path)
    if not isinstance(data, dict):
        raise ValueError(f"{ckpt_path} does not hold a checkpoint dictionary")
    sd = inference_state_dict(spec, data)
    has_ema = "state_dict" in data and data.get("ema") is not None
    if strict and not has_ema:
        known = {k for k, _, _ in model_schema(spec)}
        known |= {k[:-2] for k in known if k.endswith(".weight_v")}  # weight-norm removed
        raw = merge_lora(data["state_dict"] if "state_dict" in data else data)
        unexpected = [k for k in raw if k not in known and not k.startswith(TRAINING_ONLY_PREFIXES)]
        if unexpected:
            raise RuntimeError(f"Unexpected key(s) in state_dict: {unexpected[:5]}")
    return sd


def load_model(ckpt_path, device=None, strict=True, return_config=False, hf_token=None):
    """Load a model from a checkpoint file or a Huggingface model id `repo[:revision]`.

    Parameters are those of the reference.  `strict` follows model_loader.py:119-130: a checkpoint with an `ema` entry
    (every published one) is loaded non-strictly; otherwise `strict=True` rejects unexpected keys outside the
    training-only modules.  Missing inference tensors are fatal in either mode."""
    ckpt_path, config_path = _resolve(ckpt_path, hf_token)
    config = load_config(config_path)
    spec = spec_from_config(config)
    sd = _inference_weights(spec, ckpt_path, strict)
    cls = UniverseGAN if spec.kind == "universe_gan" else Universe
    model = cls(spec, state_dict=sd, device=device)
    model.eval()
    if return_config:
        return model, config
    return model


def load_model_sharded(ckpt_path, device=None, strict=True, return_config=False, hf_token=None):
    """`load_model` for the ranks of an initialised process group (extension; the reference is single-device): every
    rank reads the small config.yaml, ONLY rank 0 reads the checkpoint, folds and packs it, and the packed blob reaches
    the other ranks with one broadcast (RCCL over xGMI when every rank owns a GPU, `distributed.broadcast_packed_weights`)."""
    import torch.distributed as dist

    from ..distributed import broadcast_packed_weights

    ckpt_path, config_path = _resolve(ckpt_path, hf_token)
    config = load_config(config_path)
    spec = spec_from_config(config)
    rank = dist.get_rank() if dist.is_initialized() else 0
    sd, err, packed = None, None, None
    if rank == 0:
        try:  # everything that can fail on rank 0 -- reading the checkpoint AND folding / packing it -- happens before the
            # other ranks are told to enter the collective
            sd = _inference_weights(spec, ckpt_path, strict)
            from .. import _lib

            packed = _lib.pack_weights(spec, sd)[0]
        except Exception as e:
            err = e
    if dist.is_initialized():
        flag = [repr(err) if err is not None else None]
        dist.broadcast_object_list(flag, src=0)
        if flag[0] is not None:
            raise err if err is not None else RuntimeError(f"rank 0 could not load the checkpoint: {flag[0]}")
    elif err is not None:
        raise err
    blob = broadcast_packed_weights(spec, sd, device, packed=packed)
    cls = UniverseGAN if spec.kind == "universe_gan" else Universe
    model = cls(spec, packed_weights=blob, device=device)
    model.eval()
    if return_config:
        return model, config
    return model
