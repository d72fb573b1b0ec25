"""
`load_model` with the reference's signature and error behaviour
(open_universe/inference_utils/model_loader.py:62-137), returning the MI355X-native model object.

Differences, all forced by the target: the model lives on a HIP device (device=None -> current GPU; a CPU
device raises), `instantiate(config.model)` is replaced by `spec_from_config` + the C-ABI packer, and
weight-norm / EMA are resolved once at load time instead of on every forward.
"""
from pathlib import Path

import torch

from ..config import load_config, spec_from_config
from ..state_dict import inference_state_dict, model_schema
from ..universe import Universe, UniverseGAN

supported_models = ["universe"]


def ckpt_to_config_path(ckpt_path):
    """model_loader.py:33-48: <ckpt dir>/config.yaml or <ckpt dir>/../.hydra/config.yaml."""
    ckpt_path = Path(ckpt_path)
    config_path_1 = ckpt_path.parent / "config.yaml"
    config_path_2 = ckpt_path.parents[1] / ".hydra/config.yaml"
    if config_path_1.exists():
        return config_path_1
    if config_path_2.exists():
        return config_path_2
    raise ValueError(f"Could not find the configuration file for model {ckpt_path}.")


def load_model(ckpt_path, device=None, strict=True, return_config=False, hf_token=None):
    """Load a model from a checkpoint file or a Huggingface model id `repo[:revision]`.

    Parameters are those of the reference.  `strict=True` additionally rejects checkpoints that carry
    unknown (non loss/discriminator) tensors."""
    if not Path(ckpt_path).exists():
        try:
            from huggingface_hub import hf_hub_download

            ckpt_path = str(ckpt_path)
            colon_pos = ckpt_path.find(":")
            repo_id, revision = (ckpt_path[:colon_pos], ckpt_path[colon_pos + 1:]) if colon_pos >= 0 else (ckpt_path, None)
            ckpt_path = hf_hub_download(repo_id=repo_id, filename="weights.ckpt", revision=revision, token=hf_token)
            config_path = hf_hub_download(repo_id=repo_id, filename="config.yaml", revision=revision, token=hf_token)
        except Exception as e:
            print(f"{ckpt_path} is not a local file and download from HF hub failed.")
            raise e
    else:
        ckpt_path = Path(ckpt_path)
        config_path = ckpt_to_config_path(ckpt_path)

    config = load_config(config_path)
    spec = spec_from_config(config)
    data = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    sd = inference_state_dict(spec, data)
    if strict:
        known = {k for k, _, _ in model_schema(spec)}
        raw = data["state_dict"] if "state_dict" in data else data
        unexpected = [k for k in raw if k not in known and not k.startswith("loss_")]
        if unexpected:
            raise RuntimeError(f"Unexpected key(s) in state_dict: {unexpected[:5]}")
    cls = UniverseGAN if spec.kind == "universe_gan" else Universe
    model = cls(spec, state_dict=sd, device=device)
    model.eval()
    if return_config:
        return model, config
    return model
