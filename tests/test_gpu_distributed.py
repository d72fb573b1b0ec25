"""GPU (-m gpu): the N > 1 path on ONE GPU.  Two ranks share cuda:0 (RCCL refuses two ranks on one device, so the
rendezvous is gloo and the weight blob is broadcast on the host): every rank enhances its LPT shard with per-utterance
generators, and the gathered result is bit-equal to the single-process run.  Same for the CLI under two ranks and for
`bench.py --gpus 2` started WITHOUT a launcher (it must spawn its own ranks and report n_gpus = 2)."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp
import yaml

from helpers import get_spec, synth_mix
from open_universe_amd import state_dict as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENGTHS = [4100, 2900, 3555, 1600, 5200]
# batched sharded path: a RAGGED set (some lengths repeat, most groups mix lengths, one is a multiple of tot_ds = 160)
LENGTHS_B = [4100, 2900, 4100, 2900, 3555, 2900, 4100, 2880, 1600]


def _signals(spec):
    return [synth_mix(spec, 1, L, seed=60 + i)[0] for i, L in enumerate(LENGTHS)]


def _signals_b(spec):
    return [synth_mix(spec, 1, L, seed=160 + i)[0] for i, L in enumerate(LENGTHS_B)]


def _worker(rank, world, port, q, batched=False):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist

    from open_universe_amd import UniverseGAN
    from open_universe_amd import distributed as D

    r, lr, w = D.init()
    assert dist.get_world_size() == world and dist.get_backend() == "gloo"  # 1 GPU < 2 ranks
    device = D.local_device(lr)
    spec = get_spec("PP16m")
    sd = S.synthetic_state_dict(spec, seed=0) if rank == 0 else None     # only rank 0 has the checkpoint
    blob = D.broadcast_packed_weights(spec, sd, device)
    model = UniverseGAN(spec, packed_weights=blob, device=device)
    if batched:
        outs = D.enhance_sharded(model, _signals_b(spec), seed=77, n_steps=3, batch_size=4)
    else:
        outs = D.enhance_sharded(model, _signals(spec), seed=77, n_steps=3)
    if rank == 0:
        q.put([o.numpy() for o in outs])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_single_process():
    from open_universe_amd import UniverseGAN
    from open_universe_amd import distributed as D

    spec = get_spec("PP16m")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = D.free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, seed=0), device="cuda:0")
    ref = D.enhance_sharded(model, _signals(spec), seed=77, n_steps=3)  # world size 1: every utterance here
    assert len(got) == len(ref) == len(LENGTHS)
    for a, b, L in zip(got, ref, LENGTHS):
        assert a.shape == (L,) and torch.equal(torch.from_numpy(a), b)
    shards = D.shard_utterances(LENGTHS, 2)
    assert sorted(shards[0] + shards[1]) == list(range(len(LENGTHS))) and len(shards[0]) == 3  # LPT deal


def test_two_ranks_batched_shards():
    """enhance_sharded(batch_size=4) on a RAGGED set: every rank walks its LPT shard in length-sorted groups of up to four
    utterances of ANY lengths per `enhance` call -- exact batching: per-rank local max, per-row lengths (ou_enhance_var),
    per-utterance generators with the shapes of a single call.  The result of an utterance does not depend on the grouping
    or the sharding beyond fp32 summation order (the conv tilings are chosen from the total column count of a call):
    >= 100 dB against the one-call-per-utterance run, and the same grouping is bit-reproducible.  This is what each of the
    eight ranks of a node runs on its shard of a directory."""
    import restatement as O
    from helpers import record
    from open_universe_amd import UniverseGAN
    from open_universe_amd import distributed as D

    spec = get_spec("PP16m")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = D.free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, seed=0), device="cuda:0")
    sigs = _signals_b(spec)
    singles = D.enhance_sharded(model, sigs, seed=77, n_steps=3)                 # one call per utterance
    batched = D.enhance_sharded(model, sigs, seed=77, n_steps=3, batch_size=4)   # 1 rank, groups of up to 4
    again = D.enhance_sharded(model, sigs, seed=77, n_steps=3, batch_size=4)
    groups = D.plan_batches(LENGTHS_B, list(range(len(LENGTHS_B))), 4)
    assert sorted(len(g) for g in groups) == [1, 4, 4] and sum(len({LENGTHS_B[i] for i in g}) > 1 for g in groups) == 2
    for i, L in enumerate(LENGTHS_B):
        assert got[i].shape == (L,) and batched[i].shape == (L,)
        assert torch.equal(batched[i], again[i])  # same grouping: bit-identical
        record(f"sharded.batched_vs_single.{i}", O.si_sdr(singles[i].cpu(), batched[i].cpu()), 100)
        record(f"sharded.2rank_batched_vs_single.{i}", O.si_sdr(singles[i].cpu(), torch.from_numpy(got[i])), 100)
    # the older rule (equal lengths only) and exact batching agree, whatever shares a call
    eq = D.enhance_sharded(model, sigs, seed=77, n_steps=3, batch_size=4, equal_only=True)
    for i in range(len(LENGTHS_B)):
        record(f"sharded.exact_vs_equal_only.{i}", O.si_sdr(eq[i].cpu(), batched[i].cpu()), 100)
    # reference batch semantics (right zero padding, no mask): runs, keeps lengths, and is NOT the utterance alone
    padded = D.enhance_sharded(model, sigs, seed=77, n_steps=3, batch_size=4, pad_batch=True)
    assert [int(o.shape[-1]) for o in padded] == LENGTHS_B and all(torch.isfinite(o).all() for o in padded)
    assert min(float(O.si_sdr(singles[i].cpu(), padded[i].cpu())) for i in (1, 4, 8)) < 60.0


def _run(cmd, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:  # keep the whole story: pytest's assertion message truncates it
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "failed_subprocess.txt"), "a") as f:
            f.write(f"==== {' '.join(map(str, cmd))}\n---- stdout\n{r.stdout}\n---- stderr\n{r.stderr}\n")
    return r


def test_cli_under_two_ranks_equals_one_rank(tmp_path):
    """`bin/enhance.py` sharded over two processes (files dealt LPT, per-file seeds) writes the same files as one
    process with --per-file-seed."""
    from open_universe_amd import audio as A
    from open_universe_amd import config as C
    from open_universe_amd import distributed as D

    spec = get_spec("PP16s")
    mdl = tmp_path / "model"
    mdl.mkdir()
    with open(mdl / "config.yaml", "w") as f:
        yaml.safe_dump(C.builtin_config("PP16", **{"score_model.n_channels": 8}), f)
    torch.save(S.checkpoint_from_state_dict(spec, S.synthetic_state_dict(spec, seed=0), ema_jitter=0.01), mdl / "weights.ckpt")
    src = tmp_path / "in"
    (src / "sub").mkdir(parents=True)
    for i, L in enumerate([3000, 4200, 2100]):
        A.save(src / ("sub" if i == 1 else "") / f"f{i}.wav", synth_mix(spec, 1, L, seed=90 + i), 16000)
    base = [sys.executable, "-m", "open_universe_amd.bin.enhance", str(src)]
    tail = ["--model", str(mdl / "weights.ckpt"), "--n_steps", "3", "--seed", "5"]
    r1 = _run(base + [str(tmp_path / "out1")] + tail + ["--per-file-seed"])
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
               "127.0.0.1", "--master-port", str(D.free_port()), "-m", "open_universe_amd.bin.enhance", str(src),
               str(tmp_path / "out2")] + tail)
    assert r2.returncode == 0, r2.stderr[-2000:]
    for rel in ("f0.wav", "sub/f1.wav", "f2.wav"):
        a, _ = A.load(tmp_path / "out1" / rel)
        b, _ = A.load(tmp_path / "out2" / rel)
        assert torch.equal(a, b), rel


def test_cli_batch_size_matches_file_by_file(tmp_path):
    """--batch-size: consecutive files of equal rate -- of ANY lengths and channel counts -- share one enhance call (exact
    batching); the shared generator is drawn from file by file in processing order, so every file sees the noise of the
    serial loop (bin/enhance.py:147-192) and the outputs agree with the file-by-file run to fp32 summation order."""
    import restatement as O
    from helpers import record
    from open_universe_amd import audio as A
    from open_universe_amd import config as C

    spec = get_spec("PP16s")
    mdl = tmp_path / "model"
    mdl.mkdir()
    with open(mdl / "config.yaml", "w") as f:
        yaml.safe_dump(C.builtin_config("PP16", **{"score_model.n_channels": 8}), f)
    torch.save(S.checkpoint_from_state_dict(spec, S.synthetic_state_dict(spec, seed=0), ema_jitter=0.01), mdl / "weights.ckpt")
    src = tmp_path / "in"
    src.mkdir()
    lens = [3000, 3000, 3000, 4200, 4200, 2100]
    for i, L in enumerate(lens):
        A.save(src / f"f{i}.wav", synth_mix(spec, 2 if i == 1 else 1, L, seed=190 + i), 16000)  # f1 is stereo
    base = [sys.executable, "-m", "open_universe_amd.bin.enhance", str(src)]
    tail = ["--model", str(mdl / "weights.ckpt"), "--n_steps", "3", "--seed", "5"]
    r1 = _run(base + [str(tmp_path / "out1")] + tail)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = _run(base + [str(tmp_path / "out2")] + tail + ["--batch-size", "4"])
    assert r2.returncode == 0, r2.stderr[-2000:]
    for i in range(len(lens)):
        a, _ = A.load(tmp_path / "out1" / f"f{i}.wav")
        b, _ = A.load(tmp_path / "out2" / f"f{i}.wav")
        assert a.shape == b.shape
        record(f"cli.batched_vs_serial.f{i}", O.si_sdr(a.reshape(-1), b.reshape(-1)), 90)


_NCCL_WS1 = r"""
import os, sys, json
import torch, torch.distributed as dist
import yaml
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from helpers import get_spec, synth_mix
from open_universe_amd import config as C, distributed as D, state_dict as S
from open_universe_amd.inference_utils import load_model
from open_universe_amd.inference_utils.model_loader import load_model_sharded
tmp = sys.argv[2]
spec = get_spec("PP16m")
with open(os.path.join(tmp, "config.yaml"), "w") as f:
    yaml.safe_dump(C.builtin_config("PP16", **{"score_model.n_channels": 16}), f)
torch.save(S.checkpoint_from_state_dict(spec, S.synthetic_state_dict(spec, seed=0), ema_jitter=0.01), os.path.join(tmp, "weights.ckpt"))
rank, local_rank, world = D.init(backend="nccl", force=True)     # ONE rank, RCCL communicator on cuda:0
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
device = D.local_device(local_rank)
model = load_model_sharded(os.path.join(tmp, "weights.ckpt"), device=device)   # pack on rank 0 -> device-side dist.broadcast
sigs = [synth_mix(spec, 1, L, seed=60 + i)[0] for i, L in enumerate([4100, 2900, 3555])]
outs = D.enhance_sharded(model, sigs, seed=77, n_steps=3)          # gather_outputs at world 1
rep = D.rccl_report(device)
x = torch.ones(1 << 20, device=device)
dist.all_reduce(x); dist.barrier(); torch.cuda.synchronize()       # the communicator really carries a collective
assert float(x[0]) == 1.0
single = load_model(os.path.join(tmp, "weights.ckpt"), device=device)
ok = all(torch.equal(o.cpu(), single.enhance(s.to(device), n_steps=3, rng=D.utterance_generator(device, 77, i)).cpu())
         for i, (o, s) in enumerate(zip(outs, sigs)))
print("WS1_JSON " + json.dumps({"bit_equal": ok, "rccl": rep}))
dist.destroy_process_group()
"""


def test_world_size_1_nccl_group_end_to_end(tmp_path):
    """What a 1-GPU box CAN execute of the multi-GPU path: a world-size-1 `nccl` (= RCCL) process group -- communicator
    set-up with the environment RCCL needs (HSA_ENABLE_IPC_MODE_LEGACY=0), `load_model_sharded` (rank 0 packs, the blob goes
    through the DEVICE-side `dist.broadcast`), one `enhance_sharded`, an all-reduce and a barrier on the communicator --
    bit-equal to `load_model` + `enhance` without a group."""
    r = _run([sys.executable, "-c", _NCCL_WS1, ROOT, str(tmp_path)])
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("WS1_JSON ")][-1][len("WS1_JSON "):])
    assert res["bit_equal"]
    rep = res["rccl"]
    assert rep["ranks"] == 1 and rep["backend"] == "nccl" and rep["distinct_device_uuids"] == 1
    assert rep["per_rank"][0]["device"] == "cuda:0" and rep["ipc_mode_legacy"] == "0"


def test_bench_force_nccl_line_carries_the_rccl_block():
    """`bench.py --gpus 1 --force-nccl`: the headline loop under an initialised RCCL group of one rank; the JSON line carries
    `rccl: {ranks, distinct_device_uuids, backend}` (what a SCALE record needs to prove N GPUs were seen) and the
    explicit host-thread cap per rank."""
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--force-nccl", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
              "--profile-steps", "1", "--sustained-s", "0", "--in-flight", "", "--batch-sweep", ""])
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["config"]["backend"] == "nccl"
    assert res["rccl"]["ranks"] == 1 and res["rccl"]["backend"] == "nccl" and res["rccl"]["distinct_device_uuids"] == 1
    assert res["weight_broadcast"]["backend"] == "nccl" and res["weight_broadcast"]["identical_on_all_ranks"]
    assert 1 <= res["host_threads_per_rank"] <= 8


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_bench_two_gpus_over_rccl():
    """The day-one multi-GPU run: `bench.py --gpus 2` on two real devices must take the nccl (= RCCL) branch, put every
    rank on its own GPU, hand every rank rank 0's packed blob (checksum compared across ranks) and report per-rank
    times, the broadcast rate and the batch-1 / batch-4 utterance rates in one line."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
              "--profile-steps", "1"])
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["backend"] == "nccl"
    assert res["config"]["devices"] == ["rank0:cuda:0", "rank1:cuda:1"]
    assert res["rccl"]["ranks"] == 2 and res["rccl"]["distinct_device_uuids"] == 2 and res["rccl"]["backend"] == "nccl"
    wb = res["weight_broadcast"]
    assert wb["backend"] == "nccl" and wb["identical_on_all_ranks"] and wb["GBs"] > 1.0
    assert len(res["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in res["per_rank_ms_per_step"])
    assert res["batch_sweep"]["1"]["utterances_per_s"] > 0 and res["batch_sweep"]["4"]["utterances_per_s"] > 0


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: two ranks, n_gpus = 2 in the JSON line, per-rank device ids."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--share-devices", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--profile-steps", "1"])
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["devices"] == ["rank0:cuda:0", "rank1:cuda:0"]
    assert res["config"]["backend"] == "gloo" and res["value"] > 0
    assert res["weight_broadcast"]["identical_on_all_ranks"] and len(res["per_rank_ms_per_step"]) == 2
    assert set(res["batch_sweep"]) >= {"1", "4"}
