"""GPU (-m gpu): K enhance calls in flight side by side in one process (open_universe_amd.lanes.LanePool,
`enhance_sharded(..., in_flight=K)`, `bin/enhance.py --in-flight K`) -- the mode for RAGGED utterance sets, the reference
CLI's real workload (bin/enhance.py:173-192 walks a directory of files of arbitrary lengths one by one).  Every utterance is
computed by exactly the launches of the one-at-a-time call, so the outputs must be BIT-IDENTICAL to the serial loop's; what
changes is only what runs beside what (GRU clusters of all lanes resident together, one workspace per lane that serves every
length)."""
import os
import sys

import pytest
import torch

import restatement as O
from helpers import get_spec, record, synth_mix
from open_universe_amd import state_dict as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_models = {}


def get_model(name):
    from open_universe_amd import Universe, UniverseGAN

    if name not in _models:
        spec = get_spec(name)
        sd = S.synthetic_state_dict(spec, seed=0)
        cls = UniverseGAN if spec.kind == "universe_gan" else Universe
        _models[name] = (cls(spec, state_dict=sd, device="cuda:0"), spec, sd)
    return _models[name]


def ragged(spec, n, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    cand = sorted({int(v) for v in torch.randint(lo, hi, (4 * n,), generator=g).tolist()})
    lens = cand[::max(1, len(cand) // n)][:n]  # n different lengths spread over [lo, hi)
    assert len(lens) == n
    g2 = torch.Generator().manual_seed(seed + 1)
    order = torch.randperm(n, generator=g2).tolist()
    return [synth_mix(spec, 1, lens[i], seed=300 + i)[0] for i in order]


@pytest.mark.parametrize("name,n,lo,hi,lanes", [("PP16m", 12, 900, 9000, 4), ("PP16", 8, 16000, 64000, 4),
                                                ("PP16", 6, 8000, 40000, 2), ("OR16", 6, 4000, 24000, 3),
                                                ("PP24", 5, 6000, 30000, 4), ("PP16", 9, 3000, 20000, 8)])
def test_in_flight_is_bit_identical_to_the_serial_loop(name, n, lo, hi, lanes):
    from open_universe_amd import distributed as D

    model, spec, sd = get_model(name)
    sigs = ragged(spec, n, lo, hi, seed=11)
    assert len({int(s.shape[-1]) for s in sigs}) == n  # all lengths different: nothing here can be batched
    serial = D.enhance_sharded(model, sigs, seed=5, n_steps=3)
    flying = D.enhance_sharded(model, sigs, seed=5, n_steps=3, in_flight=lanes)
    again = D.enhance_sharded(model, sigs, seed=5, n_steps=3, in_flight=lanes)
    st = model.gru_exchange_stats()
    for k, (a, b, c) in enumerate(zip(serial, flying, again)):
        assert a.shape == sigs[k].shape
        # (8 lanes: two clusters per XCD -- still resident with the single call's split of the hidden units)
        assert torch.equal(a, b), (name, lanes, k, float(O.si_sdr(a, b)))
        assert torch.equal(b, c)
    assert st["lost"] == 0, st  # (a recovery of a merely LATE member may happen with lanes competing for CUs)
    # the primary model is back in single-lane mode and still agrees with itself
    solo = model.enhance(sigs[0].cuda(), n_steps=3, rng=D.utterance_generator(model.device, 5, 0)).cpu()
    assert torch.equal(solo, serial[0])


def test_in_flight_vs_oracle_and_one_workspace_for_all_lengths():
    """A lane's workspace is prepared once (ou_workspace_init) and serves every shorter length of the same batch size."""
    from open_universe_amd import distributed as D

    model, spec, sd = get_model("PP16m")
    sigs = ragged(spec, 6, 700, 5000, seed=23)
    longest = max(sigs, key=lambda s: s.shape[-1])
    model.reset_workspace()
    model.enhance(longest.cuda(), n_steps=2)  # sizes the batch-1 workspace for the longest
    ws = model._ws
    outs = D.enhance_sharded(model, sigs, seed=9, n_steps=3, in_flight=3)
    assert model._ws is ws and len(model._ws_cache) == 1
    sdict = spec.to_dict()
    for k, s in enumerate(sigs):
        # (the device generator's draws cannot be reproduced on the CPU: the oracle gets explicit noise, and so does the
        # HIP path -- on the workspace that was prepared for the longest signal)
        g = torch.Generator().manual_seed(9 + k)
        T = s.shape[-1] + (spec.tot_ds - s.shape[-1] % spec.tot_ds)
        nz = [torch.randn(1, 1, T, generator=g) for _ in range(3)]
        ref = O.enhance(sd, sdict, s[None], n_steps=3, noise=nz)[0]
        hip = model._enhance(s[None].cuda(), 3, None, None, None, None, False, False, None, "median", None,
                             [z.cuda() for z in nz]).cpu()[0]
        record(f"lanes.ws_reuse.PP16m.{k}", O.si_sdr(ref, hip), 80)
        assert outs[k].shape == s.shape


def test_growing_lengths_regrow_the_workspace():
    model, spec, sd = get_model("PP16m")
    model.reset_workspace()
    a = synth_mix(spec, 1, 1000)[0].cuda()
    b = synth_mix(spec, 1, 6000)[0].cuda()
    g = lambda: torch.Generator(device="cuda").manual_seed(3)  # noqa: E731
    ya = model.enhance(a, n_steps=2, rng=g())
    n_small = model._ws.numel()
    yb = model.enhance(b, n_steps=2, rng=g())
    assert model._ws.numel() > n_small and len(model._ws_cache) == 1
    ya2 = model.enhance(a, n_steps=2, rng=g())  # the short one again, now on the big workspace
    assert torch.equal(ya, ya2)
    model.reset_workspace()
    yb2 = model.enhance(b, n_steps=2, rng=g())
    assert torch.equal(yb, yb2)


def test_lane_stress_loop():
    """30 rounds of 4 lanes x ragged headline-size utterances: no time-out, no safety-net recovery, same bits every round."""
    from open_universe_amd import distributed as D

    model, spec, sd = get_model("PP16")
    sigs = ragged(spec, 8, 30000, 64000, seed=31)
    first = None
    for it in range(30):
        outs = D.enhance_sharded(model, sigs, seed=1, n_steps=2, in_flight=4)
        if first is None:
            first = outs
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, outs)), it
    st = model.gru_exchange_stats()
    assert st["lost"] == 0 and not model.gru_agent_scope, st


def test_cli_in_flight_matches_file_by_file(tmp_path):
    """`--in-flight K` writes the same files as the serial loop, with the shared generator (draws in processing order) and
    with --per-file-seed."""
    from open_universe_amd import audio
    from open_universe_amd.bin import enhance as cli

    model, spec, sd = get_model("PP16m")
    src = tmp_path / "in"
    src.mkdir()
    g = torch.Generator().manual_seed(4)
    for i, n in enumerate([3100, 1800, 4100, 2500, 3333, 900, 2750]):
        audio.save(src / f"f{i}.wav", 0.3 * torch.randn(1, n, generator=g).clamp(-1, 1), spec.fs)
    for extra in ([], ["--per-file-seed"]):
        o1, o2 = tmp_path / ("a" + str(len(extra))), tmp_path / ("b" + str(len(extra)))
        o1.mkdir()
        o2.mkdir()
        cli.main([str(src), str(o1), "--n_steps", "3", "--seed", "5"] + extra, model=model)
        cli.main([str(src), str(o2), "--n_steps", "3", "--seed", "5", "--in-flight", "3"] + extra, model=model)
        names = sorted(p.name for p in o1.iterdir())
        assert names == sorted(p.name for p in o2.iterdir()) and len(names) == 7
        for nme in names:
            a, fa = audio.load(o1 / nme)
            b, fb = audio.load(o2 / nme)
            assert fa == fb and torch.equal(a, b), nme


def test_enhance_many_rejects_unknown_keywords():
    """A typo must not change behaviour on the batched path only (round-3 advisor finding): `enhance_many` used to swallow
    every keyword it did not know."""
    model, spec, sd = get_model("PP16m")
    a = synth_mix(spec, 1, 1500)[0].cuda()
    with pytest.raises(TypeError, match="nsteps"):
        model.enhance_many([a, a], None, nsteps=3)
    with pytest.raises(ValueError, match="rngs"):
        model.enhance_many([a, a], None, rng=torch.Generator(device="cuda"))
    out = model.enhance_many([a, a], None, n_steps=2, ensemble_stat="median")  # (harmless: accepted)
    assert len(out) == 2 and out[0].shape == a.shape


def test_workspace_allocation_failure_releases_the_cache(monkeypatch):
    """Round-3 advisor finding: an out-of-memory on a new workspace left the idle ones cached.  Now the cache is dropped and the
    allocation retried once."""
    model, spec, sd = get_model("PP16m")
    model.reset_workspace()
    a = synth_mix(spec, 2, 1500).cuda()
    model.enhance(a, n_steps=2)           # a batch-2 workspace in the cache
    assert 2 in model._ws_cache
    real_empty, calls = torch.empty, []

    def flaky_empty(*args, **kw):
        if kw.get("dtype") is torch.uint8 and not calls:
            calls.append(1)
            raise torch.OutOfMemoryError("injected")
        return real_empty(*args, **kw)

    monkeypatch.setattr(torch, "empty", flaky_empty)
    y = model.enhance(a[:1], n_steps=2)   # needs a batch-1 workspace: first allocation fails, cache is cleared, retry succeeds
    assert calls and y.shape == (1, 1500) and list(model._ws_cache) == [1]


@pytest.mark.parametrize("kw", [dict(n_steps=4, warm_start=2), dict(n_steps=3, keep_rms=True), dict(n_steps=3, use_aux_signal=True),
                                dict(n_steps=3, ensemble=2, ensemble_stat="mean")])
def test_in_flight_with_the_cold_enhance_branches(kw):
    """warm_start / use_aux_signal (the decoupling layer's extra scratch), keep_rms and ensemble (batch of replicas per call)
    through the lanes: bit-identical to the serial loop."""
    from open_universe_amd import distributed as D

    model, spec, sd = get_model("PP16m")
    sigs = ragged(spec, 5, 1200, 6000, seed=41)
    serial = D.enhance_sharded(model, sigs, seed=2, **kw)
    flying = D.enhance_sharded(model, sigs, seed=2, in_flight=3, **kw)
    for a, b in zip(serial, flying):
        assert torch.equal(a, b)


def test_set_lanes_argument_checks_and_batched_lanes():
    """ou_set_lanes rejects nonsense; batches inside lanes (batch_size x in_flight) keep the results of the batched serial loop."""
    from open_universe_amd import distributed as D

    model, spec, sd = get_model("PP16m")
    for lanes, lane in ((0, 0), (9, 0), (2, 2), (2, -1)):
        with pytest.raises(ValueError):
            model.set_lanes(lanes, lane)
    model.set_lanes(1, 0)
    sigs = [synth_mix(spec, 1, n, seed=500 + i)[0] for i, n in enumerate([2000] * 7 + [3100] * 5 + [900])]
    serial = D.enhance_sharded(model, sigs, seed=4, n_steps=3, batch_size=4)
    flying = D.enhance_sharded(model, sigs, seed=4, n_steps=3, batch_size=4, in_flight=4)
    for a, b in zip(serial, flying):
        assert torch.equal(a, b)


@pytest.mark.parametrize("lanes", [4, 8])
def test_lanes_with_groups_of_different_batch_sizes_full_size(lanes):
    """Round-4 advisor finding: `enhance_sharded(batch_size > 1, in_flight > 1)` on a ragged set puts calls of DIFFERENT batch
    sizes in flight; shares and placement of the GRU clusters used to follow each call's own B, so on the full-size model
    (H = 256: two 8-unit clusters fill an XCD) lanes at B = 1 / 2 / 4 over-subscribed XCDs and a cluster waited for members
    that could not be scheduled -- a spurious device-side time-out.  The pool now tells every handle its largest batch size
    (ou_set_lane_batch).  Groups of 1, 2 and 4 at 4 and 8 lanes: no time-out, nothing lost, every utterance equal to the
    serial loop's to fp32 rounding (a call smaller than the pool's largest may take 16 hidden units per workgroup where the
    single call takes 8) and bit-identical from run to run."""
    from open_universe_amd import distributed as D

    model, spec, sd = get_model("PP16")
    lens = [32000] * 4 + [24000] * 2 + [28000] + [20000] * 4 + [16000] * 2 + [12000] + [36000] * 4 + [8000] * 2 + [10000]
    sigs = [synth_mix(spec, 1, n, seed=700 + i)[0] for i, n in enumerate(lens)]
    groups = D.plan_batches([int(s.shape[-1]) for s in sigs], list(range(len(sigs))), 4, equal_only=True)
    assert sorted({len(g) for g in groups}) == [1, 2, 4]  # (equal_only: the grouping that yields calls of three sizes)
    serial = D.enhance_sharded(model, sigs, seed=6, n_steps=2, batch_size=4, equal_only=True)
    flying = D.enhance_sharded(model, sigs, seed=6, n_steps=2, batch_size=4, in_flight=lanes, equal_only=True)   # raises on a time-out
    again = D.enhance_sharded(model, sigs, seed=6, n_steps=2, batch_size=4, in_flight=lanes, equal_only=True)
    st = model.gru_exchange_stats()
    assert st["lost"] == 0, st
    from helpers import worst

    record(f"lanes.mixed_batch.PP16.k{lanes}", worst(O.si_sdr(a, b) for a, b in zip(serial, flying)), 100)
    assert all(torch.equal(b, c) for b, c in zip(flying, again))
    solo = D.enhance_sharded(model, sigs[:3], seed=6, n_steps=2)  # back in single-lane mode
    assert len(solo) == 3 and model._lanes == (1, 0)


def test_graph_replay_survives_a_longer_eager_call_of_the_same_batch_size():
    """Round-4 advisor finding: workspaces are cached per batch size and regrown for a longer signal -- which used to release
    the buffer a captured graph's launches point into.  The graph now owns its workspace."""
    model, spec, sd = get_model("PP16m")
    n = 3000
    run = model.graphed_enhance(1, n, n_steps=2)
    x = synth_mix(spec, 1, n)[0].cuda()
    g = lambda: torch.Generator(device="cuda").manual_seed(12)  # noqa: E731
    y0 = run(x, rng=g())
    eager0 = model.enhance(x, n_steps=2, rng=g())
    long = model.enhance(synth_mix(spec, 1, 4 * n)[0].cuda(), n_steps=2, rng=g())  # same B, longer T: the cache regrows
    y1 = run(x, rng=g())                                                            # ... and the replay is unaffected
    assert torch.equal(y0, y1) and long.shape == (4 * n,)
    assert torch.equal(model.enhance(x, n_steps=2, rng=g()), eager0)
    model.reset_workspace()
    assert torch.equal(run(x, rng=g()), y0)  # even a cache reset leaves the graph's own buffer alone


def test_pool_close_waits_for_work_enqueued_by_a_call_that_raised():
    """Round-4 advisor finding: a callable that raised after enqueueing work left its lane marked idle, and close() handed the
    models back while kernels were still running."""
    from open_universe_amd.lanes import LanePool

    model, spec, sd = get_model("PP16m")
    x = synth_mix(spec, 1, 4000)[0].cuda()
    ref = model.enhance(x, n_steps=2, rng=torch.Generator(device="cuda").manual_seed(1))
    pool = LanePool(model, 2)

    def bad(m):
        m.enhance(x, n_steps=2, rng=torch.Generator(device="cuda").manual_seed(1))
        raise KeyError("after the enqueue")

    with pytest.raises(KeyError):
        pool.submit(bad)
    assert pool._busy[0]
    pool.close()
    assert model._lanes == (1, 0)
    assert torch.equal(model.enhance(x, n_steps=2, rng=torch.Generator(device="cuda").manual_seed(1)), ref)
    model.release_lanes()
    assert "_lane_forks" not in model.__dict__
