"""
open_universe_amd -- MI355X-native (gfx950) implementation of line/open-universe's `model.enhance` hot path.

Drop-in surface (same names / signatures as the reference):
    from open_universe_amd import inference_utils
    model = inference_utils.load_model(ckpt_path, device="cuda")
    enhanced = model.enhance(noisy)

The compute path is hand-written HIP behind a C ABI (`include/ouniverse.h`, `csrc/`); this package is the
Python host side (config, checkpoint reading, noise drawing, sharding).  There is no CPU / eager fallback.
"""
from . import config, state_dict  # noqa: F401
from . import _lib, inference_utils  # noqa: F401
from .universe import Universe, UniverseGAN  # noqa: F401

__all__ = ["config", "state_dict", "inference_utils", "Universe", "UniverseGAN"]
