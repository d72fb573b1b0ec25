"""CPU, world_size 2, gloo: the multi-GPU host path (weight broadcast, static sharding, output gather).
The data path has no collective; these are the only exchanges."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from helpers import get_spec
    from open_universe_amd import _lib
    from open_universe_amd import distributed as D
    from open_universe_amd import state_dict as S

    r, lr, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    spec = get_spec("PP16s")
    sd = S.synthetic_state_dict(spec, seed=0) if rank == 0 else None  # only rank 0 reads the checkpoint
    blob = D.broadcast_packed_weights(spec, sd, torch.device("cpu"))
    ref, _ = _lib.pack_weights(spec, S.synthetic_state_dict(spec, seed=0))
    ok = bool(torch.equal(blob, ref))
    lengths = [700, 300, 500, 100, 900]
    shards = D.shard_utterances(lengths, world)
    mine = shards[rank]
    outs = [torch.full((lengths[i],), float(i)) for i in mine]  # stand-in for the enhanced signals
    gathered = D.gather_outputs(outs, mine, len(lengths))
    if rank == 0:
        ok = ok and all(g.shape[0] == lengths[i] and float(g[0]) == i for i, g in enumerate(gathered))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(built_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


class _StubModel:
    """Stand-in with the two entry points enhance_sharded uses (CPU, no library): out = x + noise from the utterance's
    own generator -- so a wrong generator, a wrong grouping or a wrong order changes the result."""
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []

    def enhance(self, x, rng=None, **kw):
        self.calls.append(1)
        return x + torch.randn(x.shape, generator=rng)

    def enhance_many(self, sigs, rngs, pad_batch=False, **kw):
        self.calls.append(len(sigs))
        self.ragged_calls = getattr(self, "ragged_calls", 0) + int(len({int(s.shape[-1]) for s in sigs}) > 1)
        return [s + torch.randn(s.shape, generator=g) for s, g in zip(sigs, rngs)]


def _sharded_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import open_universe_amd  # noqa: F401
    from open_universe_amd import distributed as D

    D.init(backend="gloo")
    lengths = [700, 300, 700, 300, 500, 300, 700, 300, 100]
    sigs = [torch.full((n,), float(i)) for i, n in enumerate(lengths)]
    m = _StubModel()
    outs = D.enhance_sharded(m, sigs, seed=11, batch_size=4, n_steps=3)
    if rank == 0:
        q.put(([o.numpy() for o in outs], m.calls))
    dist.barrier()
    dist.destroy_process_group()


def test_enhance_sharded_batches_and_gathers_world_2():
    """enhance_sharded(batch_size=4) over two gloo ranks: LPT shards, length-sorted groups of ANY lengths per rank (exact
    batching), per-utterance generators, gather in the original order -- bit-equal to one call per utterance in one process."""
    sys.path.insert(0, ROOT)
    import open_universe_amd  # noqa: F401
    from open_universe_amd import distributed as D

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, calls = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lengths = [700, 300, 700, 300, 500, 300, 700, 300, 100]
    sigs = [torch.full((n,), float(i)) for i, n in enumerate(lengths)]
    ref = D.enhance_sharded(_StubModel(), sigs, seed=11, n_steps=3)
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert torch.equal(torch.from_numpy(a), b)
    assert max(calls) >= 2  # rank 0 really batched something
    # grouping rules
    assert D.plan_batches([5, 5, 7, 5, 7, 3], range(6), 2, equal_only=True) == [[2, 4], [0, 1], [3], [5]]
    assert D.plan_batches([5, 5, 7, 5, 7, 3], range(6), 2) == [[2, 4], [0, 1], [3, 5]]  # exact batching: any lengths
    assert D.plan_batches([5, 5, 7, 5, 7, 3], range(6), 2, pad_batch=True) == [[2, 4], [0, 1], [3, 5]]
    assert D.plan_batches([5, 5, 5], range(3), 1) == [[0], [1], [2]]


def test_enhance_sharded_empty_shard_with_calls_in_flight(monkeypatch):
    """A rank whose shard is empty (fewer utterances than ranks, or no input at all) with in_flight > 1: the pool used to be
    sized by max() over an empty set of group sizes -- ValueError on that rank while the others wait in the gather."""
    sys.path.insert(0, ROOT)
    import open_universe_amd  # noqa: F401
    from open_universe_amd import distributed as D
    from open_universe_amd import lanes

    made = []

    class FakePool:
        MAX_LANES = 8

        def __init__(self, model, n, max_batch=0):
            made.append(max_batch)
            self.m = model

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def submit(self, fn):
            return 0, fn(self.m)

        def synchronize(self):
            pass

    monkeypatch.setattr(lanes, "LanePool", FakePool)
    m = _StubModel()
    m.fork = lambda: m
    assert D.enhance_sharded(m, [], in_flight=4, batch_size=4, n_steps=3) == []
    assert made == [0]
    # ragged shard, batch_size > 1: groups of any lengths, the pool sized by the largest group
    sigs = [torch.full((n,), float(i)) for i, n in enumerate([700, 300, 500])]
    outs = D.enhance_sharded(m, sigs, seed=3, in_flight=2, batch_size=2, n_steps=3)
    ref = D.enhance_sharded(_StubModel(), sigs, seed=3, n_steps=3)
    assert all(torch.equal(a, b) for a, b in zip(outs, ref)) and made[-1] == 2 and m.ragged_calls == 1


def test_pick_backend_uses_the_local_world_size(monkeypatch):
    sys.path.insert(0, ROOT)
    import open_universe_amd  # noqa: F401
    from open_universe_amd import distributed as D

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert D.pick_backend(16) == "nccl"  # 2 nodes x 8 GPUs: one GPU per LOCAL rank
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "16")
    assert D.pick_backend(16) == "gloo"  # 16 ranks sharing 8 GPUs of one node
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert D.pick_backend(2) == "nccl" and D.pick_backend(9) == "gloo"


def _bad_ckpt_worker(rank, world, port, q, path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import open_universe_amd  # noqa: F401
    from open_universe_amd import distributed as D
    from open_universe_amd import inference_utils

    D.init(backend="gloo")
    try:
        inference_utils.load_model_sharded(path, device="cuda:0", strict=False)
        q.put((rank, "no error"))
    except Exception as e:
        q.put((rank, type(e).__name__ + ": " + str(e)[:120]))
    dist.barrier()
    dist.destroy_process_group()


def test_load_model_sharded_reports_a_packing_failure_on_every_rank(built_lib, tmp_path):
    """Only rank 0 reads and packs the checkpoint.  A tensor of the wrong shape passes the checkpoint reader and fails in the
    packer: the other ranks must hear about it BEFORE they enter the weight broadcast -- every rank raises, nobody hangs in
    the collective (round 3: the packing ran after the error-flag exchange)."""
    import yaml

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import get_spec
    from open_universe_amd import config as C
    from open_universe_amd import state_dict as S

    spec = get_spec("PP16s")
    sd = S.synthetic_state_dict(spec, seed=0)
    sd["condition_model.input_conv.bias"] = torch.zeros(3)  # wrong shape: the packer refuses it
    d = tmp_path / "m"
    d.mkdir()
    yaml.safe_dump(C.builtin_config("PP16", **{"score_model.n_channels": 8}), open(d / "config.yaml", "w"))
    torch.save(S.checkpoint_from_state_dict(spec, sd), d / "weights.ckpt")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bad_ckpt_worker, args=(r, 2, port, q, str(d / "weights.ckpt"))) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0].startswith("ValueError") and "rank 0 could not load the checkpoint" in res[1], res
