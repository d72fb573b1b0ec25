"""
Host side of the drop-in `model` object returned by `inference_utils.load_model`.

Mirrors the inference surface of the reference's `Universe` / `UniverseGAN` LightningModule
(open_universe/networks/universe/universe.py:44-386, universe_gan.py:60-149): `.fs`, `.diff_kwargs`,
`.enhance(...)` (same signature + type hints: the reference CLI introspects them,
inference_utils/signature_to_parser.py:45), `.eval()`, `.to()`, `.condition_model(...)`,
`.score_model(...)`, `.aux_to_wav(...)`.

Everything numerical happens in libouniverse.so (HIP, gfx950) through the C ABI; this class only reshapes,
draws the noise with `torch.randn(generator=rng)` in the reference's order (so a shared generator advances
identically, bin/enhance.py:147-166), owns the device buffers (PyTorch = device memory + streams) and maps C
status codes to the reference's exception types.  No torch compute fallback exists: without the library or
without a GPU construction fails.
"""
import ctypes
import math
import weakref
from ctypes import byref, c_double, c_float, c_int32, c_size_t, c_void_p
from typing import Optional

import torch

from . import _lib
from .config import ModelSpec


def randn(x, sigma, rng=None):
    """universe.py:39-41 (kept for API parity; the product path passes raw normal draws to the C ABI)."""
    noise = torch.randn(x.shape, dtype=x.dtype, device=x.device, generator=rng)
    return noise * sigma[:, None, None]


class Universe:
    """MI355X-native stand-in for the reference's `Universe` / `UniverseGAN` inference object."""

    # Steering (tests / tuning only; the library reads no environment variable): options every NEW model object starts
    # with, and the live objects -- `set_default_options` reaches both, the way an environment variable used to.
    default_options = {}
    _live = weakref.WeakSet()
    # tools/ only (each sets it explicitly): before every call, OU_<OPTION>=value environment variables are translated into
    # ou_set_option calls -- the sweep scripts of earlier rounds steer that way.  The library itself reads no environment.
    steer_from_env = False

    def __init__(self, spec: ModelSpec, state_dict=None, device=None, packed_weights=None, fir_fold=0, split_copy=True):
        """`split_copy=False`: pack / expect a blob without the bf16-split weight copy (a quarter smaller: 486 instead of 648 MB
        for UNIVERSE++ 16 kHz) -- for models that never see a batch of 8 or more utterances per call."""
        if device is None:
            device = "cuda"
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(
                f"open_universe_amd runs on MI355X (HIP) only; device={device} requested. "
                "There is no CPU path -- use the reference implementation for CPU inference."
            )
        if not torch.cuda.is_available():
            raise RuntimeError("open_universe_amd: no HIP device visible (torch.cuda.is_available() is False)")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self._L = _lib.load()  # raises LibraryNotBuilt when the extension is missing
        self.spec = spec
        self.fs = spec.fs
        self.diff_kwargs = spec.diff_kwargs
        self.normalization_norm = 2
        self.normalization_kwargs = {"ref": spec.norm_ref, "level_db": spec.level_db}
        self.with_edm = spec.edm_noise is not None
        self.tot_ds = spec.tot_ds
        self.n_channels = spec.score.n_channels
        self.device = device
        # True: synchronise the stream and raise on a device-side timeout after every call (product default).
        # False: free-running -- the status word is still copied to pinned host memory after every call (async,
        # no host sync) and examined at the start of the next call, in synchronize() and in _status(force=True).
        self.check_status = True
        self._status_host = torch.zeros(64, dtype=torch.int32).pin_memory()
        self._status_np = self._status_host.numpy()  # (same pinned memory: read per call without building tensors)
        self._gru_recoveries_seen = {}   # workspace key -> recovery counter already acted upon
        self.gru_agent_scope = False     # True once the GRU publishes were switched to agent-scope stores for good
        self._status_event = None
        self._status_ws = None
        self.training = False
        self._fir_fold = int(fir_fold)
        self._split_copy = bool(split_copy)
        self._cfg = _lib.make_config(spec, self._fir_fold, self._split_copy)
        if packed_weights is None:
            if state_dict is None:
                raise ValueError("either state_dict or packed_weights is required")
            packed_weights, _ = _lib.pack_weights(spec, state_dict, self._fir_fold, self._split_copy)
        self._weights = packed_weights.to(device=device, dtype=torch.float32).contiguous()
        self._handle = c_void_p()
        with torch.cuda.device(device):
            _lib.check(self._L.ou_create(byref(self._cfg), c_void_p(self._weights.data_ptr()),
                                         c_size_t(self._weights.numel() * 4), device.index, byref(self._handle)))
        self._ws = None
        self._ws_key = None
        self._ws_cache = {}
        self._ws_need = {}
        self._cond_key = None
        self._sigma_cache = {}
        self._lanes = (1, 0)
        for k, v in type(self).default_options.items():
            self.set_option(k, v)
        Universe._live.add(self)
        self._env_seen = None
        self._sync_env()

    def _sync_env(self):
        if not Universe.steer_from_env:
            return
        import os

        snap = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("OU_")))
        if snap == self._env_seen:
            return
        self._env_seen = snap
        env = dict(snap)
        for key, dflt in _lib.option_defaults().items():
            v = env.get("OU_" + key.upper())
            if key in ("trace", "gru_ts", "ts", "no_overlap") and v is not None:
                v = "1" if v == "" or not v.lstrip("-").isdigit() else v  # (these used to be presence flags)
            self.set_option(key, dflt if v is None else float(v))
        _lib.check(self._L.ou_set_stamp_layer(self._handle, env.get("OU_CHAIN_TS", "").encode()), self._handle)

    # ---- steering through the C ABI (ou_set_option): tests force kernel families, tools sweep; never needed for normal use ----
    def set_option(self, key, value):
        """One typed option of this model's handle (keys: `_lib.option_names()`); takes effect from the next call on."""
        _lib.check(self._L.ou_set_option(self._handle, str(key).encode(), float(value)), self._handle)
        self._ws_need.clear()  # (options may change what a walk allocates)

    def get_option(self, key):
        v = c_double()
        _lib.check(self._L.ou_get_option(self._handle, str(key).encode(), byref(v)), self._handle)
        return v.value

    def options(self):
        return {k: self.get_option(k) for k in _lib.option_names()}

    def reset_options(self):
        _lib.check(self._L.ou_reset_options(self._handle), self._handle)
        self._ws_need.clear()

    @classmethod
    def set_default_options(cls, **kw):
        """Set options on every live model object AND on those created from now on (value None: back to the default)."""
        defaults = _lib.option_defaults()
        for k, v in kw.items():
            if v is None:
                Universe.default_options.pop(k, None)
            else:
                Universe.default_options[k] = v
            for m in list(Universe._live):
                m.set_option(k, defaults[k] if v is None else v)

    # ---- several enhance calls in flight in one process -----------------------------------------------------------
    def fork(self):
        """A second model object on the SAME packed weights (no copy) with a handle, workspace and status record of its
        own: what one lane of `LanePool` runs on.  A handle is not re-entrant; several handles side by side are fine."""
        twin = type(self)(self.spec, packed_weights=self._weights, device=self.device, fir_fold=self._fir_fold,
                          split_copy=self._split_copy)
        twin.check_status = self.check_status
        for k, v in self.options().items():  # a lane runs what its primary model would run
            twin.set_option(k, v)
        if self.gru_agent_scope:
            twin.gru_agent_scope = True
            _lib.check(twin._L.ou_set_gru_publish_mode(twin._handle, 1), twin._handle)
        return twin

    def set_lanes(self, lanes, lane, max_batch=0):
        """This object is lane `lane` of `lanes` models whose calls are in flight side by side on this device (one stream
        each): the library then sizes and places the GRU clusters of every lane so that all of them fit on the device
        together (include/ouniverse.h, ou_set_lanes).  `max_batch`: the largest batch size ANY lane of the pool will run
        when the calls differ in size (ou_set_lane_batch; 0 = every call's own size)."""
        _lib.check(self._L.ou_set_lanes(self._handle, int(lanes), int(lane)), self._handle)
        _lib.check(self._L.ou_set_lane_batch(self._handle, int(max_batch)), self._handle)
        self._lanes = (int(lanes), int(lane))

    def release_lanes(self):
        """Drop the forks (handles + workspaces) that `LanePool`s of this model created and kept for re-use."""
        for twin in self.__dict__.pop("_lane_forks", []):
            twin.reset_workspace()

    # ------------------------------------------------------------------------------------------------
    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None and self._handle.value:
                self._L.ou_destroy(self._handle)
                self._handle = c_void_p()
        except Exception:
            pass

    def eval(self, no_ema=False):
        """universe.py:864-865: inference always runs on the EMA weights (resolved at load time)."""
        return self

    def train(self, mode=True, no_ema=False):
        if mode:
            raise NotImplementedError("open_universe_amd implements the inference (enhance) path only")
        return self

    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args else None)
        if dev is not None and torch.device(dev).type != "cuda":
            raise RuntimeError("open_universe_amd models live on a HIP device; .to(cpu) is not supported")
        return self

    # ------------------------------------------------------------------------------------------------
    def _stream(self):
        return c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # workspaces kept alive, ONE per batch size (most recently used last): a buffer prepared by ou_workspace_init for
    # (B, T) serves every shorter length of the same batch size (include/ouniverse.h), so a directory of files of different
    # lengths runs on one buffer that grows to the longest file seen; a workspace costs an allocation plus the init
    WS_CACHE_ENTRIES = 4
    WS_CACHE_BYTES = 16 << 30

    def _workspace_bytes(self, B, T):
        n = self._ws_need.get((B, T))
        if n is None:
            c = c_size_t()
            _lib.check(self._L.ou_workspace_bytes(self._handle, B, T, byref(c)), self._handle)
            n = self._ws_need[(B, T)] = c.value
            if len(self._ws_need) > 4096:
                self._ws_need.clear()
        return n

    def _workspace(self, B, T):
        key = (B, T)
        if self._ws_key == key:
            return self._ws
        need = self._workspace_bytes(B, T)
        cache = self._ws_cache
        ws = cache.get(B)
        if ws is None or ws.numel() < need:
            # free-running mode: the status copy of the previous call belongs to the workspace that is left now -- the next
            # call's copy would replace it unexamined
            if self._status_event is not None and not self.check_status:
                self._status_event.synchronize()
                self._raise_on_status()
            cache.pop(B, None)
            while cache and (len(cache) >= self.WS_CACHE_ENTRIES
                             or sum(w.numel() for w in cache.values()) + need > self.WS_CACHE_BYTES):
                cache.pop(next(iter(cache)))
            self._ws = ws = None
            try:
                ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            except torch.OutOfMemoryError:
                # give the idle workspaces (and the allocator's cached blocks) back and try once more
                cache.clear()
                torch.cuda.empty_cache()
                ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(self._L.ou_workspace_init(self._handle, B, T, c_void_p(ws.data_ptr()),
                                                     c_size_t(need), self._stream()), self._handle)
        elif self._ws is not ws and self._status_event is not None and not self.check_status:
            self._status_event.synchronize()
            self._raise_on_status()
        cache.pop(B, None)
        cache[B] = ws  # most recently used last
        self._ws = ws
        self._ws_key = key
        self._cond_key = None
        return self._ws

    def _private_workspace(self, B, T):
        """A workspace of its own for a captured graph: never in the cache, so no eager call -- same batch size, longer
        signal -- can regrow or evict the buffer the graph's launches point into."""
        need = self._workspace_bytes(B, T)
        ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ou_workspace_init(self._handle, B, T, c_void_p(ws.data_ptr()), c_size_t(need), self._stream()),
                       self._handle)
        return ws

    def _adopt_workspace(self, ws, B, T):
        """Make `ws` (prepared for at least (B, T)) the current workspace of this object."""
        if self._ws is ws and self._ws_key == (B, T):
            return
        if self._ws is not ws and self._status_event is not None and not self.check_status:
            # free-running mode: the pending status copy belongs to the workspace that is left now
            self._status_event.synchronize()
            self._raise_on_status()
        self._ws = ws
        self._ws_key = (B, T)
        self._cond_key = None

    def reset_workspace(self):
        """Drop every cached workspace: the next call allocates and initialises a fresh one (tests; after switching
        between kernel generations that lay the GRU exchange area out differently)."""
        if self._status_event is not None:
            # a status copy of an earlier call may still be pending: look at it while its workspace is still known
            self._status_event.synchronize()
            try:
                self._raise_on_status()
            finally:
                self._status_event = None
        self._ws_cache.clear()
        self._ws = None
        self._ws_key = None
        self._cond_key = None
        self._status_ws = None

    def _raise_on_status(self):
        self._status_event = None
        v = int(self._status_np[0])
        # word 20: waits the GRU clusters' safety net cut short by repeating a publish (0 on an idle device; a member that was
        # merely late -- starved by the kernels of other streams or lanes -- counts too); word 33: those among them where the
        # awaited granule was visible to a system-scope load / an atomic but not to the agent-scope load of the gather.  The
        # first time word 33 moves, the cheaper publish form has shown that it cannot be relied upon on this device / in this
        # process mix: switch to agent-scope (write-through) publishes for good -- +0.1 ms per GRU pass, no more recoveries.
        rec, lost = int(self._status_np[20]), int(self._status_np[33])
        if lost and not self.gru_agent_scope and not v:
            import warnings

            self.gru_agent_scope = True
            _lib.check(self._L.ou_set_gru_publish_mode(self._handle, 1), self._handle)
            warnings.warn(f"open_universe_amd: {lost} of {rec} repeated GRU hand-off(s) were publishes that an agent-scope load "
                          f"did not see (first event: {self._status_host[21:30].tolist()}); results are unaffected, the "
                          "recurrence kernels publish with agent-scope stores from now on", RuntimeWarning)
        if v:
            ws = self._status_ws if self._status_ws is not None else self._ws
            diag = ws[:256].view(torch.int32).cpu().tolist()  # who waited for what (see gru_ring_kernel)
            self._status_host.zero_()
            ws[:4].zero_()  # the device word is sticky until cleared
            ws[32 * 4:60 * 4].zero_()
            # [12..19]: reporting cluster / member / step / min tag seen / tag wanted / XCC / plain stores / block id;
            # [32] max tag seen, [36..]: (step << 8 | xcc now << 4 | xcc at the rendezvous) of every member of that cluster
            # that was itself stuck in a long wait (0xFFFFFFFF: slot not of this launch); [21..29]: first-recovery record
            members = [("-" if m == -1 else f"{(m & 0xFFFFFFFF) >> 8}@x{m & 0xFF:02x}") for m in diag[36:60]]
            if v & 16 and not v & ~16:
                # bit 16: the barrier of the fused deep-ConvBlock launch (conv_block3_kernel, OU_BLOCK3=1 only) ran out
                raise RuntimeError("device-side timeout in the fused ConvBlock launch's group barrier (status word 16, "
                                   "OU_BLOCK3=1); the output of that call is invalid and the workspace has to be "
                                   "re-initialised (Universe.reset_workspace())")
            raise RuntimeError(f"device-side timeout in the GRU cluster exchange (status word {v}, diagnostics "
                               f"{diag[8:20]}, max tag {diag[32] & 0xFFFFFFFF}, first recovery record {diag[21:30]}, "
                               f"members' waits {members}); the output of that call is invalid")

    def _poll_deferred_status(self):
        """Free-running mode: look at the status copy of an EARLIER call once its event has completed."""
        if self._status_event is not None and self._status_event.query():
            self._raise_on_status()

    def _status(self, force=False):
        """Called after every forward: enqueue the (async) copy of the device status word; in the default mode -- or
        with force=True -- wait for it and raise if a kernel flagged a timeout."""
        if self._ws is None:
            return
        st = torch.cuda.current_stream(self.device)
        if self._status_ws is not self._ws:
            self._status_ws = self._ws
            self._status_words = self._ws[:256].view(torch.int32)
        self._status_host.copy_(self._status_words, non_blocking=True)  # (on the current stream, behind the call's last kernel)
        ev = torch.cuda.Event()
        ev.record(st)
        self._status_event = ev
        if self.check_status or force:
            ev.synchronize()
            self._raise_on_status()

    def synchronize(self):
        """Wait for everything enqueued by this model and raise if any call since the last check timed out."""
        torch.cuda.current_stream(self.device).synchronize()
        if self._status_event is not None:
            self._raise_on_status()

    def gru_exchange_stats(self):
        """Health of the GRU clusters' L2 hand-offs on the current workspace: `recoveries` = publishes that had to be
        repeated by the safety net (each costs ~0.1 ms), `system_scope` = waves that finished their GRU pass with
        system-scope publishes after such a recovery (slower steps, no more recoveries).  Both 0 on a healthy device."""
        if self._ws is None:
            return {"recoveries": 0, "lost": 0, "system_scope": 0}
        d = self._ws[:256].view(torch.int32).cpu().tolist()
        # recoveries: waits cut short by the safety net (late members included); lost: publishes that really were invisible
        out = {"recoveries": int(d[20]), "lost": int(d[33]), "system_scope": int(d[31])}
        out["agent_scope_publishes"] = bool(self.gru_agent_scope)
        if d[21]:  # first recovery on this workspace: what three kinds of loads saw in the stale granule (gru_stale_probe)
            out["first_event"] = {"events": d[21], "cluster": (d[22] >> 16) & 0xFFFF, "member": (d[22] >> 8) & 0xFF,
                                  "wave": d[22] & 0xFF, "granule": d[23], "want": d[24], "sc1": d[25], "sc0sc1": d[26],
                                  "atomic": d[27], "sc1_after_inv": d[28], "step": (d[29] >> 16) & 0xFFFF,
                                  "xcc_at_rendezvous": (d[29] >> 8) & 0xFF, "xcc_now": d[29] & 0xFF}
        return out

    def tensor(self, name):
        """Debug: view of a named intermediate of the last call inside the workspace -> (B, C, T) tensor."""
        off, C, T = c_size_t(), c_int32(), c_int32()
        rc = self._L.ou_tensor(self._handle, name.encode(), byref(off), byref(C), byref(T))
        if rc != 0:
            raise KeyError(name)
        B = self._ws_key[0]
        n = B * C.value * T.value
        return self._ws[off.value: off.value + 4 * n].view(torch.float32).view(B, C.value, T.value)

    def launch_stats(self):
        a, b = c_int32(), c_int32()
        self._L.ou_launch_stats(self._handle, byref(a), byref(b))
        return a.value, b.value

    def profile(self, on):
        """Bracket every generic-conv launch of the following calls with HIP events (measurement only)."""
        _lib.check(self._L.ou_profile_enable(self._handle, int(bool(on))), self._handle)

    def profile_read(self, max_records=8192):
        """-> list of (ms, algorithmic_flops, algorithmic_bytes, tile_cfg) per conv launch since profile(True)."""
        ms = (c_float * max_records)()
        fl = (ctypes.c_double * max_records)()
        by = (ctypes.c_double * max_records)()
        cf = (c_int32 * max_records)()
        n = c_int32()
        _lib.check(self._L.ou_profile_read(self._handle, max_records, ms, fl, by, cf, byref(n)), self._handle)
        return [(ms[i], fl[i], by[i], cf[i]) for i in range(n.value)]

    def profile_read_ticks(self, max_records=32768):
        """-> list of (start, end, variant) per profiled launch: 10 ns ticks of the device's constant clock."""
        t0 = (ctypes.c_uint64 * max_records)()
        t1 = (ctypes.c_uint64 * max_records)()
        cf = (c_int32 * max_records)()
        n = c_int32()
        _lib.check(self._L.ou_profile_read_ticks(self._handle, max_records, t0, t1, cf, byref(n)), self._handle)
        return [(t0[i], t1[i], cf[i]) for i in range(n.value)]

    def bench_conv(self, layer, B, Tin, cfg=-1, sc=-1, with_res=False, iters=20):
        """Tuning aid: ms per launch of one packed conv layer (see ou_bench_conv)."""
        self._sync_env()
        ws = torch.empty(max(1 << 28, 64 * B * Tin * 4 * 64), dtype=torch.uint8, device=self.device)
        ms, used = c_float(), c_int32()
        _lib.check(self._L.ou_bench_conv(self._handle, layer.encode(), B, Tin, cfg, sc, int(with_res), iters,
                                         c_void_p(ws.data_ptr()), c_size_t(ws.numel()), self._stream(), byref(ms),
                                         byref(used)), self._handle)
        return ms.value, used.value

    def pad(self, x, pad=None):
        """universe.py:219-223."""
        if pad is None:
            pad = self.tot_ds - x.shape[-1] % self.tot_ds
        return torch.nn.functional.pad(x, (pad // 2, pad - pad // 2)), pad

    def unpad(self, x, pad):
        return x[..., pad // 2: -(pad - pad // 2)]

    def get_std_dev(self, time):
        """universe.py:380-386 (geometric schedule)."""
        s_min, s_max = self.diff_kwargs.sigma_min, self.diff_kwargs.sigma_max
        return s_min * (s_max / s_min) ** time

    # ---- operator seams (universe.py:314-316, 286) --------------------------------------------------
    def _prep(self, x):
        if x.device != self.device:
            raise ValueError(f"input is on {x.device}, model on {self.device}")
        return x.to(torch.float32).contiguous()

    def condition_model(self, x, x_wav=None, train=False):
        """condition.py:346-377.  x: (B,1,T) normalised, T % tot_ds == 0.  Returns conditions or, with
        train=True, (conditions, aux_signal, latent) -- views into the workspace, valid until the next call."""
        self._sync_env()
        x = self._prep(x)
        if x.ndim != 3 or x.shape[1] != 1:
            raise ValueError("condition_model expects (B, 1, T)")
        B, _, T = x.shape
        if T % self.tot_ds:
            raise ValueError("the HIP conditioner needs T % tot_ds == 0 (enhance() pads accordingly)")
        ws = self._workspace(B, T)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ou_condition(self._handle, c_void_p(x.data_ptr()), B, T, c_void_p(ws.data_ptr()),
                                            c_size_t(ws.numel()), self._stream()), self._handle)
        self._cond_key = (B, T)
        self._status()
        n_blocks = len(self.spec.score.rate_factors) + int(self.spec.cond.extra_conv_block)
        cond = [self.tensor(f"cond.c{j}") for j in range(n_blocks)]
        if train:
            return cond, self.tensor("cond.aux"), self.tensor("cond.latent")
        return cond

    def score_model(self, x, sigma, cond=None):
        """universe.py:197-209 / score.py:277-297: score(x, sigma | cond of the last condition_model call)."""
        self._sync_env()
        x = self._prep(x)
        B, _, T = x.shape
        if self._cond_key != (B, T):
            raise ValueError("score_model: call condition_model on a (B,1,T) input of the same shape first")
        sig = sigma.detach().to(torch.float32).cpu().contiguous()
        if sig.numel() != B:
            raise ValueError("sigma must have one entry per batch element")
        out = torch.empty_like(x)
        ws = self._ws
        with torch.cuda.device(self.device):
            _lib.check(self._L.ou_score(self._handle, c_void_p(x.data_ptr()),
                                        ctypes.cast(sig.data_ptr(), ctypes.POINTER(c_float)), c_void_p(out.data_ptr()),
                                        B, T, c_void_p(ws.data_ptr()), c_size_t(ws.numel()), self._stream()), self._handle)
        self._status()
        return out

    def aux_to_wav(self, y_aux=None):
        """universe_gan.py:145-149 on the aux signal of the last condition_model call."""
        if not self.spec.use_signal_decoupling:
            if y_aux is None:
                return self.tensor("cond.aux")
            return y_aux
        B, T = self._cond_key
        out = torch.empty(B, 1, T, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ou_aux_to_wav(self._handle, c_void_p(out.data_ptr()), B, T, c_void_p(self._ws.data_ptr()),
                                             c_size_t(self._ws.numel()), self._stream()), self._handle)
        self._status()
        return out

    # ---- the hot path ------------------------------------------------------------------------------
    def enhance(
        self,
        mix,
        n_steps: Optional[int] = None,
        epsilon: Optional[float] = None,
        target: Optional[torch.Tensor] = None,
        fake_score_snr: Optional[float] = None,
        rng: Optional[torch.Generator] = None,
        use_aux_signal: Optional[bool] = False,
        keep_rms: Optional[bool] = False,
        ensemble: Optional[int] = None,
        ensemble_stat: Optional[str] = "median",
        warm_start: Optional[int] = None,
    ) -> torch.Tensor:
        """Universe.enhance, universe.py:231-375 (same arguments, same return convention)."""
        return self._enhance(mix, n_steps, epsilon, target, fake_score_snr, rng, use_aux_signal, keep_rms, ensemble,
                             ensemble_stat, warm_start, None)

    @torch.no_grad()
    def _enhance(self, mix, n_steps, epsilon, target, fake_score_snr, rng, use_aux_signal, keep_rms, ensemble,
                 ensemble_stat, warm_start, noise, t_raw=None):
        """`t_raw`: per-row lengths of a batch whose rows are utterances of different lengths (enhance_many, exact
        batching -> ou_enhance_var); `mix` is then (B, 1, max length) and `noise` a (n, B, 1, T) tensor."""
        self._sync_env()
        self._poll_deferred_status()
        if epsilon is None:
            epsilon = self.diff_kwargs.epsilon
        if n_steps is None:
            n_steps = self.diff_kwargs.n_steps
        x_ndim = mix.ndim
        if x_ndim == 1:
            mix = mix[None, None, :]
        elif x_ndim == 2:
            mix = mix[:, None, :]
        elif x_ndim > 3:
            raise ValueError("The input should have at most 3 dimensions")
        if mix.ndim == 3 and mix.shape[1] != 1:
            raise ValueError("enhance expects single-channel signals: (T,), (B,T) or (B,1,T)")
        if ensemble_stat not in ("mean", "median", "signal_median") and ensemble is not None:
            raise NotImplementedError()  # universe.py:368
        mix = self._prep(mix)
        if ensemble is not None:
            mix_shape = mix.shape
            if keep_rms and mix_shape[0] != 1:
                # universe.py:259 computes mix_rms before the replication (:261-264): the reference fails to
                # broadcast (B,1,1) against (E*B,1,T) at :354 for B > 1; same behaviour here.
                raise RuntimeError("keep_rms with ensemble is only defined for a single input signal (as in the reference)")
            mix = torch.stack([mix] * ensemble, dim=0).view((-1,) + mix_shape[1:])
        B, _, mix_len = mix.shape
        pad = self.tot_ds - mix_len % self.tot_ds
        T = mix_len + pad

        if target is not None:
            x = self._enhance_with_oracle_score(mix, target, n_steps, epsilon, fake_score_snr, rng, pad)
        else:
            n_start = 0 if warm_start is None else int(warm_start)
            n_noise = 0 if use_aux_signal else n_steps - n_start
            if t_raw is not None:
                if ensemble is not None:
                    raise ValueError("per-row lengths and `ensemble` do not combine (call enhance per input)")
                noise_t = noise
                if n_noise and (noise_t is None or tuple(noise_t.shape) != (n_noise, B, 1, T)):
                    raise ValueError(f"noise must be a tensor of shape {(n_noise, B, 1, T)}")
            elif noise is None:
                # draw order of the reference: x0, then z_n for n = n_start .. N-2 (universe.py:326,330,338)
                # (each draw lands in its slice of the tensor the C ABI takes: the same n separate (B, 1, T) draws -- generator
                # state and values are those of the reference's loop -- without a stack copy per draw behind them)
                # (n_noise <= 0: warm_start >= n_steps -- the C ABI refuses the call below)
                noise_t = torch.empty((n_noise, B, 1, T), dtype=torch.float32, device=self.device) if n_noise > 0 else None
                for k in range(n_noise):
                    torch.randn((B, 1, T), generator=rng, out=noise_t[k])
            elif torch.is_tensor(noise):  # (all steps in one (n, B, 1, T) tensor: taken as it is)
                noise_t = self._prep(noise[:n_noise]) if n_noise else None
                if n_noise and tuple(noise_t.shape) != (n_noise, B, 1, T):
                    raise ValueError(f"noise must be a tensor of shape {(n_noise, B, 1, T)}")
            else:
                noise_t = torch.stack([self._prep(z) for z in noise[:n_noise]], dim=0) if n_noise else None
                if n_noise and noise_t.shape != (n_noise, B, 1, T):
                    raise ValueError(f"noise must be {n_noise} tensors of shape {(B, 1, T)}")
            # discretised schedule exactly as the reference builds it (universe.py:308-311) -- host arithmetic, done once per
            # (n_steps, schedule) and AFTER the draws are in the queue: the device is idle while the host prepares a call
            skey = (int(n_steps), float(self.diff_kwargs.sigma_min), float(self.diff_kwargs.sigma_max))
            sigma = self._sigma_cache.get(skey)
            if sigma is None:
                time = torch.linspace(0, 1, n_steps).to(torch.float32).flip(dims=[0])
                sigma = self.get_std_dev(time).to(torch.float32).contiguous()
                if len(self._sigma_cache) > 64:
                    self._sigma_cache.clear()
                self._sigma_cache[skey] = sigma
            out = torch.empty(B, 1, mix_len, dtype=torch.float32, device=self.device)
            ws = self._workspace(B, T)
            flags = (_lib.OU_ENH_KEEP_RMS if keep_rms else 0) | (_lib.OU_ENH_USE_AUX_SIGNAL if use_aux_signal else 0)
            with torch.cuda.device(self.device):
                if t_raw is not None:
                    if len(t_raw) != B:
                        raise ValueError("t_raw must have one entry per row of the batch")
                    rows_len = (c_int32 * B)(*[int(v) for v in t_raw])
                    _lib.check(self._L.ou_enhance_var(
                        self._handle, c_void_p(mix.data_ptr()), c_void_p(out.data_ptr()),
                        c_void_p(noise_t.data_ptr()) if noise_t is not None else None, B, mix_len, rows_len, int(n_steps),
                        float(epsilon), ctypes.cast(sigma.data_ptr(), ctypes.POINTER(c_float)),
                        -1 if warm_start is None else int(warm_start), flags, c_void_p(ws.data_ptr()),
                        c_size_t(ws.numel()), self._stream()), self._handle)
                else:
                    _lib.check(self._L.ou_enhance(
                        self._handle, c_void_p(mix.data_ptr()), c_void_p(out.data_ptr()),
                        c_void_p(noise_t.data_ptr()) if noise_t is not None else None, B, mix_len, int(n_steps),
                        float(epsilon), ctypes.cast(sigma.data_ptr(), ctypes.POINTER(c_float)),
                        -1 if warm_start is None else int(warm_start), flags, c_void_p(ws.data_ptr()),
                        c_size_t(ws.numel()), self._stream()), self._handle)
            self._cond_key = (B, T) if t_raw is None else None
            self._status()
            x = out

        if target is not None:
            if keep_rms:
                mix_rms = mix.square().mean(dim=(-2, -1), keepdim=True).sqrt()
                x_rms = x.square().mean(dim=(-2, -1), keepdim=True).sqrt().clamp(min=1e-5)
                x = x * (mix_rms / x_rms)
            scale = abs(x).max(dim=-1, keepdim=True).values
            x = torch.where(scale > 1.0, x / scale, x)

        if ensemble is not None:  # universe.py:359-368 (host-side glue over E replicas of the same utterance)
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

    @torch.no_grad()
    def enhance_many(self, signals, rngs=None, pad_batch=False, n_steps=None, epsilon=None, use_aux_signal=False,
                     keep_rms=False, warm_start=None, **other):
        """Several independent inputs in ONE `enhance` call (extension; the reference's CLI loops over files one by one,
        bin/enhance.py:173-192).  `signals`: list of (L,) or (C, L) tensors -- a (C, L) entry is a file whose channels
        are rows of the batch, as in the reference.  `rngs`: one generator per entry, ONE shared generator, or None.
        The noise of entry i is drawn from its generator entry by entry, step by step, with the shapes a call on that
        entry alone would use ((C_i, 1, T), x0 first) -- with a shared generator the draws come in exactly the order
        of the serial loop, so its state advances as the reference's does.
        pad_batch=False (default): EXACT batching -- entries may have any lengths, and every row is the signal it would be in
        a call of its own (own pad() split, own normalisation and mel norm, zero padding of every conv right behind its own
        last sample, GRU passes over its own frames: ou_enhance_var); the noise of entry i has the shape of that call,
        (C_i, 1, L_i + pad_i).  Agrees with the one-by-one loop to fp32 round-off (the kernels a batch selects differ).
        pad_batch=True: right-zero-padded to the longest entry like `max_collator` (datasets/datamodule.py:24-42);
        the reference has no mask, the padding takes part in the normalisation / mel norm / GRU; outputs are cropped.
        Returns the list of enhanced signals, each with the shape of its input."""
        for k in ("target", "ensemble", "fake_score_snr"):
            if other.get(k) is not None:
                raise ValueError(f"enhance_many does not take `{k}` (call enhance per input)")
        unknown = set(other) - {"target", "ensemble", "fake_score_snr", "ensemble_stat", "rng"}
        if unknown:  # (a typo must not change behaviour on the batched path only)
            raise TypeError(f"enhance_many() got unexpected keyword argument(s): {sorted(unknown)}")
        if other.get("rng") is not None:
            raise ValueError("enhance_many takes the generators as `rngs` (one per input, or one shared)")
        if not signals:
            return []
        rows, dims = [], []
        for s in signals:
            if s.ndim not in (1, 2):
                raise ValueError("enhance_many takes (L,) or (C, L) signals")
            dims.append(s.ndim)
            rows.append(self._prep(s if s.ndim == 2 else s[None, :]))
        lens = [int(r.shape[-1]) for r in rows]
        if min(lens) < 1:
            raise ValueError("enhance_many: empty input signal")
        l_max = max(lens)
        n_steps = self.diff_kwargs.n_steps if n_steps is None else int(n_steps)
        T = l_max + (self.tot_ds - l_max % self.tot_ds)
        n_start = 0 if warm_start is None else int(warm_start)
        if n_start >= n_steps:
            raise ValueError("warm_start must be < n_steps")
        n_noise = 0 if use_aux_signal else n_steps - n_start
        if not pad_batch and any(n != l_max for n in lens):
            # exact batching of different lengths: per-row geometry through the whole path
            B = sum(r.shape[0] for r in rows)
            noise_t = torch.zeros((n_noise, B, 1, T), dtype=torch.float32, device=self.device) if n_noise else None
            t_raw, r0 = [], 0
            for i, (r, n) in enumerate(zip(rows, lens)):
                g = rngs[i] if isinstance(rngs, (list, tuple)) else rngs
                Ti = n + (self.tot_ds - n % self.tot_ds)
                for k in range(n_noise):  # the draws of the call on this entry alone, in its order (x0 first)
                    dst = noise_t[k, r0:r0 + r.shape[0], :, :Ti]
                    if dst.is_contiguous():  # (one row: the draw goes straight to its place)
                        torch.randn((r.shape[0], 1, Ti), generator=g, out=dst)
                    else:
                        dst.copy_(torch.randn((r.shape[0], 1, Ti), dtype=torch.float32, device=self.device, generator=g))
                t_raw += [n] * r.shape[0]
                r0 += r.shape[0]
            mix = torch.cat([torch.nn.functional.pad(r, (0, l_max - r.shape[-1])) for r in rows], dim=0)[:, None, :]
            out = self._enhance(mix, n_steps, epsilon, None, None, None, use_aux_signal, keep_rms, None, "median",
                                warm_start, noise_t, t_raw=t_raw)
            res, r0 = [], 0
            for r, nd, n in zip(rows, dims, lens):
                o = out[r0:r0 + r.shape[0], 0, :n]
                r0 += r.shape[0]
                res.append(o[0] if nd == 1 else o)
            return res
        # entry by entry, step by step (the serial loop's draw order), every draw straight into its rows of the step's tensor
        B = sum(r.shape[0] for r in rows)
        noise = torch.empty((n_noise, B, 1, T), dtype=torch.float32, device=self.device)
        r0 = 0
        for i, r in enumerate(rows):
            g = rngs[i] if isinstance(rngs, (list, tuple)) else rngs
            for k in range(n_noise):
                torch.randn((r.shape[0], 1, T), generator=g, out=noise[k, r0:r0 + r.shape[0]])
            r0 += r.shape[0]
        mix = torch.cat([torch.nn.functional.pad(r, (0, l_max - r.shape[-1])) for r in rows], dim=0)[:, None, :]
        out = self._enhance(mix, n_steps, epsilon, None, None, None, use_aux_signal, keep_rms, None, "median",
                            warm_start, noise)
        res, r0 = [], 0
        for r, nd, n in zip(rows, dims, lens):
            o = out[r0:r0 + r.shape[0], 0, :n]
            r0 += r.shape[0]
            res.append(o[0] if nd == 1 else o)
        return res

    def advance_generator_like_enhance(self, rng, channels, length, n_steps=None, warm_start=None, use_aux_signal=False):
        """Advance `rng` by exactly the draws `enhance` makes for a (channels, length) input -- x0, then one z per noisy step
        (universe.py:326,330,338), each of shape (channels, 1, length + pad) -- without running anything.  For callers that
        re-order work but owe every input the noise of the serial loop: take `rng.get_state()` in front of an input, call this,
        and hand a generator restored to that state to the call that really processes the input (the CLI's length-sorted
        window, bin/enhance.py)."""
        n_steps = self.diff_kwargs.n_steps if n_steps is None else int(n_steps)
        n_noise = 0 if use_aux_signal else n_steps - (0 if warm_start is None else int(warm_start))
        T = int(length) + (self.tot_ds - int(length) % self.tot_ds)
        for _ in range(n_noise):
            torch.randn((int(channels), 1, T), dtype=torch.float32, device=self.device, generator=rng)

    # ---- hipGraph replay of the hot path ------------------------------------------------------------------------
    def graphed_enhance(self, batch, length, n_steps=None, epsilon=None, keep_rms=False, serial=True):
        """-> callable `run(mix, rng=None)` equivalent to `enhance(mix, n_steps, epsilon, rng=rng, keep_rms=keep_rms)`
        for inputs of shape (batch, length): ONE `ou_enhance` (pad .. peak guard, ~430 launches over the caller's stream
        and three side streams) is captured into a hipGraph once and replayed per call -- the host then enqueues one
        graph launch instead of walking the network (the GPU timeline is the same back-to-back sequence either way).
        The noise is still drawn per call with `torch.randn(generator=rng)` in the reference's order, into the static
        buffer the graph reads, so a shared generator advances exactly as in the eager path.  The library call is
        capturable by construction: no allocation, no host synchronisation, GRU exchange tags advance on the device."""
        n_steps = self.diff_kwargs.n_steps if n_steps is None else int(n_steps)
        epsilon = self.diff_kwargs.epsilon if epsilon is None else float(epsilon)
        if not serial:
            # Measured (tools/hwq_probe.sh, profiles/r05_final_hwq_probe.txt): with GPU_MAX_HW_QUEUES below 4 the HIP runtime
            # SEGFAULTS replaying a captured graph that has this call's four-stream fork / join structure (round 4 recorded it
            # as a hang of bench.py); the eager call and the serial-chain capture are fine with 1, 2 or 4 queues.
            import os

            hwq = os.environ.get("GPU_MAX_HW_QUEUES")
            if hwq is not None and hwq.strip().isdigit() and int(hwq) < 4:
                raise RuntimeError(f"graphed_enhance(serial=False) captures four streams; GPU_MAX_HW_QUEUES={hwq} makes the HIP "
                                   "runtime crash when such a graph is replayed -- use serial=True (the default)")
        B, mix_len = int(batch), int(length)
        T = mix_len + (self.tot_ds - mix_len % self.tot_ds)
        dev = self.device
        s_mix = torch.zeros(B, 1, mix_len, dtype=torch.float32, device=dev)
        s_noise = torch.zeros(n_steps, B, 1, T, dtype=torch.float32, device=dev)
        s_out = torch.empty(B, 1, mix_len, dtype=torch.float32, device=dev)
        sigma = self.get_std_dev(torch.linspace(0, 1, n_steps).to(torch.float32).flip(dims=[0])).to(torch.float32).contiguous()
        # the graph's launches carry this buffer's address: a workspace of the graph's own, outside the per-batch-size cache
        # (an eager call with the same B and a longer T regrows the cached one)
        ws = self._private_workspace(B, T)
        self._adopt_workspace(ws, B, T)
        # serial=True: capture the call as ONE chain on the capture stream.  The eager path forks three side streams inside
        # the call (mel branch, st convs, first score-encoder pass); captured, every fork / join becomes a cross-stream edge
        # of the graph, and such a graph replays SLOWER than the chain (measured, profiles/).
        flags = (_lib.OU_ENH_KEEP_RMS if keep_rms else 0) | (_lib.OU_ENH_SERIAL if serial else 0)

        def launch():
            _lib.check(self._L.ou_enhance(
                self._handle, c_void_p(s_mix.data_ptr()), c_void_p(s_out.data_ptr()), c_void_p(s_noise.data_ptr()), B,
                mix_len, n_steps, epsilon, ctypes.cast(sigma.data_ptr(), ctypes.POINTER(c_float)), -1, flags,
                c_void_p(ws.data_ptr()), c_size_t(ws.numel()), self._stream()), self._handle)

        with torch.cuda.device(dev):
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                launch()  # warm-up outside the capture (first-touch work, kernel attribute setup)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                launch()
        self._cond_key = (B, T)

        def run(mix, rng=None):
            self._poll_deferred_status()
            x = self._prep(mix).reshape(B, 1, mix_len)
            self._adopt_workspace(ws, B, T)  # (eager calls on any shape in between are fine: the graph owns its workspace)
            s_mix.copy_(x)
            for n in range(n_steps):  # draw order of the reference: x0, z_0 .. z_{N-2}
                s_noise[n].copy_(torch.randn((B, 1, T), dtype=torch.float32, device=dev, generator=rng))
            graph.replay()
            self._status()
            out = s_out.clone()
            if mix.ndim == 1:
                return out[0, 0]
            if mix.ndim == 2:
                return out[:, 0, :]
            return out

        run.graph = graph
        return run

    def _enhance_with_oracle_score(self, mix, target, n_steps, epsilon, fake_score_snr, rng, pad):
        """Diagnostic mode of the reference (universe.py:278-298): the network is bypassed by the analytic
        score of a known target, so no kernel of this library is involved -- plain device tensor glue."""
        tot = self.tot_ds
        level = 10 ** (self.spec.level_db / 20.0)

        def stats(t):
            mean = t.mean(dim=(1, 2), keepdim=True)
            return mean, level / (t - mean).std(dim=(1, 2), keepdim=True).clamp(min=1e-5)

        def padded(t):
            return torch.nn.functional.pad(t, (pad // 2, pad - pad // 2))

        # utils/norm.py:47-87: ref == "both" normalises the target by its own statistics, "noisy" by the mixture's
        mixp, tgt = padded(mix), padded(self._prep(target))
        m_mean, m_gain = stats(mixp)
        t_mean, t_gain = stats(tgt) if self.normalization_kwargs["ref"] == "both" else (m_mean, m_gain)
        mixp = (mixp - m_mean) * m_gain
        tgt = (tgt - t_mean) * t_gain
        score_snr = 5.0 if fake_score_snr is None else fake_score_snr
        delta_t = 1.0 / (n_steps - 1)
        gamma = (self.diff_kwargs.sigma_max / self.diff_kwargs.sigma_min) ** -delta_t
        eta = 1 - gamma ** epsilon
        beta = math.sqrt(1 - gamma ** (2 * (epsilon - 1.0)))
        time = torch.linspace(0, 1, n_steps).type_as(mixp).flip(dims=[0])
        sigma = self.get_std_dev(time)
        sigma = torch.broadcast_to(sigma[None, :], (mixp.shape[0], sigma.shape[0]))

        def score_wrapper(x, s):
            true_score = -(x - tgt) / s[:, None, None] ** 2
            noise_rms = (true_score ** 2).mean().sqrt() * 10 ** (-score_snr / 20.0)
            nz = torch.randn(true_score.shape, dtype=true_score.dtype, device=true_score.device, generator=rng)
            return true_score + nz * noise_rms

        x = randn(mixp, sigma[:, 0], rng=rng)
        for n in range(n_steps - 1):
            s_now, s_next = sigma[:, n], sigma[:, n + 1]
            score = score_wrapper(x, s_now)
            z = randn(x, s_next, rng=rng)
            x = x + s_now[..., None, None] ** 2 * eta * score + beta * z
        x = x + sigma[:, -1, None, None] ** 2 * score_wrapper(x, sigma[:, -1])
        x = x[..., pad // 2: -(pad - pad // 2)]
        return torch.nn.functional.pad(x, (0, mix.shape[-1] - x.shape[-1]))


class UniverseGAN(Universe):
    """UNIVERSE++ (universe_gan.py:60): same inference surface; aux_to_wav goes through the decoupling layer."""


def signal_median(signal):
    """utils/stats.py:22-66 semantics without the sort: for every (batch entry, sample) the reference looks up, in the
    ascending order of the n ensemble members, the POSITION of the member whose index is nearest to n/2 (first hit in
    rank order on a tie), histograms those positions per batch entry and returns member number argmax(histogram).
    Here the position of member c is obtained by counting the members that precede it ("x_j < x_c", ties broken by
    member index like a stable sort), and the histogram is one bincount over (batch entry, position) pairs."""
    shape = signal.shape
    x = signal.flatten(start_dim=2)  # (n, B, S)
    n, B, S = x.shape
    dist = (torch.arange(n, dtype=torch.float64) - n / 2).abs()
    cands = [c for c in range(n) if float(dist[c]) == float(dist.min())]
    pos = None
    for c in cands:
        before = (x < x[c]).sum(dim=0)
        if c > 0:
            before = before + (x[:c] == x[c]).sum(dim=0)
        pos = before if pos is None else torch.minimum(pos, before)  # (B, S): rank position, smaller wins a tie
    rows = torch.arange(B, device=x.device)[:, None] * n
    hist = torch.bincount((rows + pos).flatten(), minlength=B * n).view(B, n)
    pick = hist.argmax(dim=1)  # first maximum, like the reference's counts.argmax
    out = x[pick, torch.arange(B, device=x.device)]
    return out.reshape(shape[1:])
