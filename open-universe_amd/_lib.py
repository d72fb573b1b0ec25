"""
ctypes binding of libouniverse.so (the C ABI declared in include/ouniverse.h).

This is the only bridge between the Python host side and the HIP kernels.  There is no fallback: if the
shared library is missing (not built) every entry point raises, loudly.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint32, c_void_p

OU_MAX_RATES = 8
OU_ABI_VERSION = 5
OU_OK, OU_EINVAL, OU_ENOTIMPL, OU_EMISSING, OU_ESHAPE, OU_EHIP, OU_ENOMEM, OU_ESYNC = 0, -1, -2, -3, -4, -5, -6, -7
OU_KIND_UNIVERSE, OU_KIND_UNIVERSE_GAN = 0, 1
OU_ACT_NONE, OU_ACT_PRELU, OU_ACT_SNAKE = 0, 1, 2
OU_ENH_KEEP_RMS, OU_ENH_USE_AUX_SIGNAL, OU_ENH_NO_PEAK_GUARD, OU_ENH_SERIAL = 1, 2, 4, 8


class NetConfig(Structure):
    _fields_ = [
        ("n_rates", c_int32),
        ("rate_factors", c_int32 * OU_MAX_RATES),
        ("n_channels", c_int32),
        ("fb_kernel_size", c_int32),
        ("n_rff", c_int32),
        ("noise_cond_dim", c_int32),
        ("extra_conv_block", c_int32),
        ("use_weight_norm", c_int32),
        ("use_antialiasing", c_int32),
        ("time_embedding_simple", c_int32),
        ("n_mels", c_int32),
        ("n_mel_oversample", c_int32),
        ("encoder_gru_residual", c_int32),
    ]


class Config(Structure):
    _fields_ = [
        ("abi_version", c_int32),
        ("kind", c_int32),
        ("fs", c_int32),
        ("level_db", c_float),
        ("has_edm", c_int32),
        ("edm_noise", c_float),
        ("sigma_min", c_double),
        ("sigma_max", c_double),
        ("use_signal_decoupling", c_int32),
        ("signal_decoupling_act", c_int32),
        ("score", NetConfig),
        ("cond", NetConfig),
        ("has_edm_data_level", c_int32),
        ("edm_data_level_db", c_float),
        ("fir_fold", c_int32),
        ("no_split_copy", c_int32),
    ]


class LibraryNotBuilt(RuntimeError):
    pass


_LIB = None


def lib_path():
    return os.environ.get("OU_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libouniverse.so"))


def load():
    """Load libouniverse.so.  Raises LibraryNotBuilt (never falls back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise LibraryNotBuilt(
            f"{path} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C open-universe_amd/csrc`).  open_universe_amd has no CPU / eager fallback."
        )
    L = ctypes.CDLL(path)
    vp, sz, i32 = c_void_p, c_size_t, c_int32
    sig = {
        "ou_version": (c_char_p, []),
        "ou_last_error": (c_char_p, [vp]),
        "ou_packer_last_error": (c_char_p, [vp]),
        "ou_packer_create": (i32, [POINTER(Config), POINTER(vp)]),
        "ou_packer_set": (i32, [vp, c_char_p, vp, POINTER(c_int64), i32]),
        "ou_packer_finish": (i32, [vp, POINTER(vp), POINTER(sz)]),
        "ou_packer_destroy": (None, [vp]),
        "ou_packed_bytes": (i32, [POINTER(Config), POINTER(sz)]),
        "ou_create": (i32, [POINTER(Config), vp, sz, i32, POINTER(vp)]),
        "ou_destroy": (None, [vp]),
        "ou_workspace_bytes": (i32, [vp, i32, i32, POINTER(sz)]),
        "ou_schedule": (i32, [POINTER(Config), i32, c_double, POINTER(c_float), POINTER(c_double), POINTER(c_double)]),
        "ou_condition": (i32, [vp, vp, i32, i32, vp, sz, vp]),
        "ou_score": (i32, [vp, vp, POINTER(c_float), vp, i32, i32, vp, sz, vp]),
        "ou_aux_to_wav": (i32, [vp, vp, i32, i32, vp, sz, vp]),
        "ou_enhance": (i32, [vp, vp, vp, vp, i32, i32, i32, c_double, POINTER(c_float), i32, c_uint32, vp, sz, vp]),
        "ou_enhance_var": (i32, [vp, vp, vp, vp, i32, i32, POINTER(i32), i32, c_double, POINTER(c_float), i32, c_uint32, vp, sz, vp]),
        "ou_check_device_status": (i32, [vp, vp]),
        "ou_set_option": (i32, [vp, c_char_p, c_double]),
        "ou_get_option": (i32, [vp, c_char_p, POINTER(c_double)]),
        "ou_reset_options": (i32, [vp]),
        "ou_option_count": (i32, []),
        "ou_option_name": (c_char_p, [i32]),
        "ou_option_doc": (c_char_p, [i32]),
        "ou_option_default": (c_double, [i32]),
        "ou_set_stamp_layer": (i32, [vp, c_char_p]),
        "ou_plan_json": (c_char_p, [vp]),
        "ou_packer_plan_json": (c_char_p, [vp]),
        "ou_tensor": (i32, [vp, c_char_p, POINTER(sz), POINTER(i32), POINTER(i32)]),
        "ou_launch_stats": (i32, [vp, POINTER(i32), POINTER(i32)]),
        "ou_workspace_init": (i32, [vp, i32, i32, vp, sz, vp]),
        "ou_sampler_step": (i32, [vp, vp, vp, vp, c_float, c_float, sz, vp]),
        "ou_set_gru_publish_mode": (i32, [vp, i32]),
        "ou_get_gru_publish_mode": (i32, [vp]),
        "ou_set_lanes": (i32, [vp, i32, i32]),
        "ou_set_lane_batch": (i32, [vp, i32]),
        "ou_lane_capacity": (i32, [vp, i32]),
        "ou_transform_frames": (i32, [i32, i32, i32]),
        "ou_transform_forward": (i32, [vp, i32, i32, vp, i32, i32, i32, c_float, c_float, vp, vp]),
        "ou_transform_inverse": (i32, [vp, i32, i32, vp, i32, i32, i32, c_float, c_float, i32, vp, vp, vp]),
        "ou_flac_last_error": (c_char_p, []),
        "ou_flac_info": (i32, [vp, sz, POINTER(i32), POINTER(i32), POINTER(i32), POINTER(ctypes.c_int64), vp]),
        "ou_flac_decode": (i32, [vp, sz, vp, ctypes.c_int64, POINTER(ctypes.c_int64)]),
        "ou_profile_enable": (i32, [vp, i32]),
        "ou_bench_conv": (i32, [vp, c_char_p, i32, i32, i32, i32, i32, i32, vp, sz, vp, POINTER(c_float), POINTER(i32)]),
        "ou_profile_read": (i32, [vp, i32, POINTER(c_float), POINTER(c_double), POINTER(c_double), POINTER(i32), POINTER(i32)]),
        "ou_profile_read_ticks": (i32, [vp, i32, POINTER(ctypes.c_uint64), POINTER(ctypes.c_uint64), POINTER(i32), POINTER(i32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


EXPORTED_SYMBOLS = [
    "ou_version", "ou_last_error", "ou_packer_last_error", "ou_packer_create", "ou_packer_set", "ou_packer_finish",
    "ou_packer_destroy", "ou_packed_bytes", "ou_create", "ou_destroy", "ou_workspace_bytes", "ou_schedule",
    "ou_condition", "ou_score", "ou_aux_to_wav", "ou_enhance", "ou_enhance_var", "ou_check_device_status",
    "ou_set_option", "ou_get_option", "ou_reset_options", "ou_option_count", "ou_option_name", "ou_option_doc", "ou_option_default", "ou_plan_json",
    "ou_packer_plan_json", "ou_tensor", "ou_launch_stats", "ou_workspace_init", "ou_sampler_step",
    "ou_set_gru_publish_mode", "ou_get_gru_publish_mode", "ou_set_lanes", "ou_set_lane_batch", "ou_lane_capacity",
    "ou_transform_frames", "ou_transform_forward", "ou_transform_inverse",
    "ou_flac_last_error", "ou_flac_info", "ou_flac_decode",
]
TUNING_SYMBOLS = ["ou_profile_enable", "ou_profile_read", "ou_profile_read_ticks", "ou_bench_conv", "ou_set_stamp_layer",
]

_EXC = {OU_EINVAL: ValueError, OU_ENOTIMPL: NotImplementedError, OU_EMISSING: KeyError, OU_ESHAPE: ValueError,
        OU_EHIP: RuntimeError, OU_ENOMEM: MemoryError, OU_ESYNC: RuntimeError}


def check(code, handle=None, packer=None):
    """Map a C status code to the exception type the reference raises for the same condition."""
    if code == OU_OK:
        return
    L = load()
    if packer is not None:
        msg = L.ou_packer_last_error(packer)
    else:
        msg = L.ou_last_error(handle)
    msg = msg.decode() if msg else f"libouniverse error {code}"
    raise _EXC.get(code, RuntimeError)(msg)


def option_names():
    """Keys of ou_set_option, with their one-line descriptions."""
    L = load()
    return {L.ou_option_name(i).decode(): L.ou_option_doc(i).decode() for i in range(L.ou_option_count())}


def option_defaults():
    L = load()
    return {L.ou_option_name(i).decode(): L.ou_option_default(i) for i in range(L.ou_option_count())}


def make_config(spec, fir_fold=0, split_copy=True):
    """ModelSpec -> ou_config.  `fir_fold`, `split_copy`: packing choices (ou_config.fir_fold / .no_split_copy), the same for the
    packer and ou_create."""
    def net(n):
        c = NetConfig()
        c.n_rates = len(n.rate_factors)
        for i, r in enumerate(n.rate_factors):
            c.rate_factors[i] = int(r)
        c.n_channels = n.n_channels
        c.fb_kernel_size = n.fb_kernel_size
        c.n_rff = n.n_rff
        c.noise_cond_dim = n.noise_cond_dim
        c.extra_conv_block = int(n.extra_conv_block)
        c.use_weight_norm = int(n.use_weight_norm)
        c.use_antialiasing = int(n.use_antialiasing)
        c.time_embedding_simple = int(n.time_embedding == "simple")
        c.n_mels = n.n_mels
        c.n_mel_oversample = n.n_mel_oversample
        c.encoder_gru_residual = int(n.encoder_gru_residual)
        return c

    cfg = Config()
    cfg.abi_version = OU_ABI_VERSION
    cfg.kind = OU_KIND_UNIVERSE_GAN if spec.kind == "universe_gan" else OU_KIND_UNIVERSE
    cfg.fs = spec.fs
    cfg.level_db = spec.level_db
    cfg.has_edm = int(spec.edm_noise is not None)
    cfg.edm_noise = spec.edm_noise if spec.edm_noise is not None else 0.0
    cfg.sigma_min = spec.sigma_min
    cfg.sigma_max = spec.sigma_max
    cfg.use_signal_decoupling = int(spec.use_signal_decoupling)
    cfg.signal_decoupling_act = {"snake": OU_ACT_SNAKE, "prelu": OU_ACT_PRELU}.get(spec.signal_decoupling_act, OU_ACT_NONE)
    cfg.score = net(spec.score)
    cfg.cond = net(spec.cond)
    data_level = getattr(spec, "edm_data_level_db", None)
    cfg.has_edm_data_level = int(data_level is not None)
    cfg.edm_data_level_db = float(data_level) if data_level is not None else 0.0
    cfg.fir_fold = int(fir_fold)
    cfg.no_split_copy = 0 if split_copy else 1
    return cfg


def pack_weights(spec, state_dict, fir_fold=0, split_copy=True):
    """Fold + lay out a reference-keyed state dict into the device blob (host side, no GPU needed).
    Returns (torch.FloatTensor blob [CPU], plan_json str)."""
    import numpy as np
    import torch

    L = load()
    cfg = make_config(spec, fir_fold, split_copy)
    packer = c_void_p()
    check(L.ou_packer_create(byref(cfg), byref(packer)))
    try:
        for key, t in state_dict.items():
            t = t.detach().to(torch.float32).cpu().contiguous()
            shape = (c_int64 * max(1, t.ndim))(*t.shape)
            check(L.ou_packer_set(packer, key.encode(), c_void_p(t.data_ptr()), shape, t.ndim), packer=packer)
        blob_p, nbytes = c_void_p(), c_size_t()
        check(L.ou_packer_finish(packer, byref(blob_p), byref(nbytes)), packer=packer)
        arr = np.ctypeslib.as_array(ctypes.cast(blob_p, POINTER(c_float)), shape=(nbytes.value // 4,))
        blob = torch.from_numpy(arr.copy())
        plan = L.ou_packer_plan_json(packer).decode()
    finally:
        L.ou_packer_destroy(packer)
    return blob, plan


def packed_bytes(spec, fir_fold=0, split_copy=True):
    L = load()
    cfg = make_config(spec, fir_fold, split_copy)
    n = c_size_t()
    check(L.ou_packed_bytes(byref(cfg), byref(n)))
    return n.value
