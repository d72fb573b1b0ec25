"""
Single-node multi-GPU driver for the enhance path: one process per GPU, utterances sharded statically, packed
weights distributed once by an RCCL broadcast over xGMI, NO collective inside the sampling loop
(utterances are fully independent -- normalize_batch and the mel norm reduce per sample;
the ensemble reduce stays on one rank).

The reference has no multi-GPU inference (bin/enhance.py:173-192 is a serial per-file loop on one device).

Backends: `nccl` (= RCCL) when every rank owns its own GPU; `gloo` otherwise -- CPU-only hosts (tests) and the
"several ranks share one GPU" arrangement used to exercise the N > 1 path on a 1-GPU box: RCCL refuses two ranks on one
device, so the weight blob is then broadcast over gloo on the host and copied to the device by each rank.
"""
import os
import socket
import subprocess
import sys

# dmabuf IPC only on these hosts (RCCL needs it).  ROCr reads the variable when the runtime loads, i.e. at the first HIP call
# of the process -- so it is set HERE, at import, before anything of this package can have touched the device; whether that
# was early enough is recorded (a process that initialised HIP before importing this module keeps what it had).
_IPC_PRESET = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

_IPC_SET_BEFORE_HIP_INIT = _IPC_PRESET is not None or not torch.cuda.is_initialized()

from . import _lib  # noqa: E402


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def pick_backend(world=None):
    """nccl (RCCL) iff there is one visible GPU per LOCAL rank; gloo for CPU hosts and shared-device runs.
    `device_count()` is per node, so it is held against LOCAL_WORLD_SIZE (torchrun sets it; a multi-node job with
    8 ranks per node and world = 16 is still one GPU per local rank), falling back to the global size."""
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world if world is not None else os.environ.get("WORLD_SIZE", "1")))
    if torch.cuda.is_available() and torch.cuda.device_count() >= local_world:
        return "nccl"
    return "gloo"


def local_device(local_rank):
    """The HIP device of this rank (ranks wrap around the visible devices when they share GPUs), or CPU."""
    if not torch.cuda.is_available():
        return torch.device("cpu")
    return torch.device("cuda", local_rank % torch.cuda.device_count())


def init(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment (nccl == RCCL on ROCm; gloo on CPU / shared GPU).
    A single process normally runs without a process group; `force=True` creates one anyway (world size 1, rendezvous on
    127.0.0.1) so that communicator set-up, the environment RCCL needs and the device-side weight broadcast execute on a
    1-GPU box exactly as they do on a node (`bench.py --force-nccl`, tests/test_gpu_distributed.py)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = pick_backend(world)
        if torch.cuda.is_available():
            torch.cuda.set_device(local_device(local_rank))
        if not _IPC_SET_BEFORE_HIP_INIT and world > 1 and backend == "nccl":
            import warnings

            warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY was not in the environment when HIP was initialised in this process; "
                          "export HSA_ENABLE_IPC_MODE_LEGACY=0 before the launcher if RCCL fails with hipIpcGetMemHandle errors",
                          RuntimeWarning)
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())
        kw = {}
        if backend == "nccl":  # bind the communicator to this rank's device up front (no lazy guess at the first collective)
            kw["device_id"] = local_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def cap_host_threads(local_world=None):
    """One process per GPU: the ranks of a node share its host cores.  Caps torch's intra-op pool of THIS process at
    min(8, logical CPUs / local ranks) unless OMP_NUM_THREADS says otherwise -- the enhance path needs one enqueueing thread;
    eight uncapped ranks would start 8 x 256 OpenMP threads for the few host-side tensor ops there are."""
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if "OMP_NUM_THREADS" in os.environ:
        n = max(1, int(os.environ["OMP_NUM_THREADS"]))
    else:
        n = max(1, min(8, (os.cpu_count() or 8) // max(1, local_world)))
    torch.set_num_threads(n)
    return n


def rccl_report(device):
    """What a multi-GPU record has to prove: how many ranks there were, that each sat on its OWN physical GPU (distinct
    device UUIDs, not just distinct ordinals) and which backend carried the collectives.  All ranks must call it."""
    if not dist.is_initialized():
        return None
    device = torch.device(device)
    me = {"rank": dist.get_rank(), "device": str(device), "uuid": None, "name": None, "pci": None,
          "host_threads": torch.get_num_threads()}
    if device.type == "cuda":
        pr = torch.cuda.get_device_properties(device)
        me["uuid"] = str(getattr(pr, "uuid", "")) or None
        me["name"] = pr.name
        me["pci"] = f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}"
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, me)
    ids = [e["uuid"] or e["pci"] or e["device"] for e in every]
    ver = None
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    return {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "distinct_device_uuids": len(set(ids)),
            "rccl_version": ver, "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "ipc_mode_set_before_hip_init": bool(_IPC_SET_BEFORE_HIP_INIT),
            "per_rank": every}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn_under_launcher(script, argv, nproc, timeout=None):
    """`python <script> --gpus N` started WITHOUT a launcher: re-execute the same command line under
    `torch.distributed.run` with N local ranks (rendezvous on 127.0.0.1) and return its exit code.  stdout / stderr are
    inherited, so the one JSON line rank 0 prints is the output of the outer process too."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


def shard_utterances(lengths, world_size):
    """Static LPT partition: sort by length (descending), deal round-robin.  Returns, per rank, the list of
    utterance indices it owns (each rank then pads only to its local maximum)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards = [[] for _ in range(world_size)]
    for k, idx in enumerate(order):
        shards[k % world_size].append(idx)
    return shards


def broadcast_packed_weights(spec, state_dict, device, src=0, packed=None):
    """Rank `src` folds + packs the checkpoint (or hands in the blob it has already packed: `packed`); everybody receives
    the blob with ONE broadcast (PP16: 185 MB, one xGMI hop).  Returns the device tensor to hand to
    Universe(packed_weights=...).  Packing can fail (missing / mis-shaped tensors): a caller whose other ranks are already
    waiting in the collective should pack first, tell them, and pass `packed` (inference_utils.load_model_sharded)."""
    grouped = dist.is_initialized()
    world = dist.get_world_size() if grouped else 1
    rank = dist.get_rank() if grouped else 0
    device = torch.device(device)
    nfloats = _lib.packed_bytes(spec) // 4
    # the collective runs where the backend lives: on the GPUs for RCCL, on the host for gloo
    on_host = grouped and dist.get_backend() == "gloo"
    xdev = torch.device("cpu") if on_host else device
    if rank == src:
        blob = packed if packed is not None else _lib.pack_weights(spec, state_dict)[0]
        blob = blob.to(xdev)
    else:
        blob = torch.empty(nfloats, dtype=torch.float32, device=xdev)
    if grouped:  # (also with ONE rank when a group exists: the same code path as on a node, cheap)
        dist.broadcast(blob, src=src)
    return blob.to(device)


def gather_outputs(local_outputs, local_indices, n_total, dst=0):
    """Optional: collect enhanced signals on `dst` (list of 1-D CPU tensors in the original order)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    payload = [(int(i), o.detach().cpu()) for i, o in zip(local_indices, local_outputs)]
    if world == 1:
        out = [None] * n_total
        for i, o in payload:
            out[i] = o
        return out
    gathered = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if dist.get_rank() != dst:
        return None
    out = [None] * n_total
    for part in gathered:
        for i, o in part:
            out[i] = o
    return out


def utterance_generator(device, seed, index):
    """Per-utterance noise stream: generator of utterance k is seeded with seed + k, so that the enhanced signal of an
    utterance does not depend on which rank -- or beside which other utterances -- it was processed."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) + int(index))
    return g


def plan_batches(lengths, indices, batch_size, pad_batch=False, equal_only=False):
    """Group the utterances `indices` (of raw lengths `lengths[i]`) into `enhance` calls of at most `batch_size` rows.

    Neighbours in length order share a call (longest first: the padding a batch wastes is the difference to its longest
    member).  pad_batch=False (default): EXACT batching -- every row keeps its own geometry (`Universe.enhance_many`,
    ou_enhance_var) and is the utterance it would be alone, whatever it shares the call with.  pad_batch=True: the
    reference's batch semantics for ragged sets (`max_collator`, datasets/datamodule.py:24-42): right-zero-padded to the
    longest of the call without a mask, so the padding is seen by the normalisation, the mel norm and the GRU (SURVEY.md
    8(d), C5), and a row is NOT the utterance it would be alone.  equal_only=True: only utterances of EQUAL raw length
    share a call (the rule before exact batching existed; a call then never carries per-row lengths).
    Deterministic: a pure function of the arguments."""
    batch_size = max(1, int(batch_size))
    order = sorted(indices, key=lambda i: (-int(lengths[i]), i))
    groups = []
    if pad_batch or not equal_only:
        for k in range(0, len(order), batch_size):
            groups.append(order[k:k + batch_size])
        return groups
    cur = []
    for i in order:
        if cur and (int(lengths[i]) != int(lengths[cur[0]]) or len(cur) == batch_size):
            groups.append(cur)
            cur = []
        cur.append(i)
    if cur:
        groups.append(cur)
    return groups


def enhance_sharded(model, signals, seed=1028282, gather=True, batch_size=1, pad_batch=False, in_flight=1,
                    equal_only=False, **enhance_kwargs):
    """Enhance a list of 1-D signals (any lengths) across the ranks of the current process group.

    Rank r takes its LPT shard (`shard_utterances`) and walks it in `plan_batches` groups: up to `batch_size`
    utterances per `enhance` call (one call per utterance with the default batch_size=1).  Every utterance draws its
    noise from its own generator (`utterance_generator`, seed + index) with the shapes a call on that utterance alone
    would use, so with pad_batch=False the result of an utterance does not depend on the sharding or the grouping
    beyond fp32 summation order (the conv tilings are chosen from the total column count of a call: batched vs single
    agree to > 100 dB, same grouping = bit-identical; a 1-rank and an N-rank run with batch_size=1 are bit-identical).
    Utterances of DIFFERENT lengths share a call too (exact batching: every row keeps its own padding, statistics, conv
    zero padding and GRU length, `Universe.enhance_many`); `equal_only=True` restores the older rule (equal lengths only).

    in_flight=K > 1: K calls of the shard are in flight side by side on K streams of this rank's GPU (`lanes.LanePool`:
    one handle + workspace per lane): every call is exactly the call of the serial loop (same kernels, tilings, summation
    orders), so the outputs are bit-identical to in_flight=1 while the device overlaps the GRU passes and the small
    launches of one utterance with the kernels of the others (batch_size=1: the route for ragged sets before exact
    batching, 143 -> 238 utt/s; batching reaches the batched kernels' rate instead).

    Results are gathered on rank 0 in the original order (None on the other ranks; with gather=False every rank
    returns {index: tensor} of its shard)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lengths = [int(s.shape[-1]) for s in signals]
    mine = shard_utterances(lengths, world)[rank]
    outs = {}
    batched_ok = not any(enhance_kwargs.get(k) is not None for k in ("target", "ensemble"))
    groups = plan_batches(lengths, mine, batch_size if batched_ok else 1, pad_batch, equal_only)

    def run_group(m, group):
        if len(group) == 1:
            i = group[0]
            return [m.enhance(signals[i].to(m.device), rng=utterance_generator(m.device, seed, i), **enhance_kwargs)]
        return m.enhance_many([signals[i].to(m.device) for i in group],
                              [utterance_generator(m.device, seed, i) for i in group], pad_batch=pad_batch,
                              **enhance_kwargs)

    in_flight = max(1, int(in_flight))
    if in_flight > 1 and getattr(model, "fork", None) is not None and enhance_kwargs.get("target") is None:
        from .lanes import LanePool

        sizes = {len(g) for g in groups}
        # (groups of different sizes in flight side by side: the lanes must agree on the GRU cluster layout -- the pool's largest;
        #  a rank whose shard is EMPTY -- fewer utterances than ranks -- has no group to size the pool by: 0 = every call's own)
        mb = max(sizes, default=0)
        with LanePool(model, min(in_flight, LanePool.MAX_LANES), max_batch=mb if (len(sizes) > 1 or mb > 1) else 0) as pool:
            for group in groups:
                _, res = pool.submit(lambda m, g=group: run_group(m, g))
                for i, o in zip(group, res):
                    outs[i] = o
            pool.synchronize()  # raises on a device-side time-out of any call of the shard
    else:
        # The shard runs free: no host synchronisation between the calls (the device status word is sticky and is examined
        # once after the loop), so the host work of the next call -- generator seeding, the walk of the network, ~400
        # launches -- overlaps the kernels of the current one.  With a sync after every call that work sat between the
        # calls: 8.27 instead of 7.4 ms per utterance at batch_size=1.
        sync_mode = getattr(model, "check_status", None)
        if sync_mode is not None:
            model.check_status = False
        try:
            for group in groups:
                for i, o in zip(group, run_group(model, group)):
                    outs[i] = o
        finally:
            if sync_mode is not None:
                model.check_status = sync_mode
        if sync_mode is not None:
            model.synchronize()  # raises on a device-side time-out of any call of the shard
    if not gather:
        return outs
    return gather_outputs([outs[i] for i in mine], mine, len(signals))
