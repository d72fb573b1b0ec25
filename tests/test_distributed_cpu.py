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
