"""GPU (-m gpu): the two generations of the GRU cluster kernel (torch.nn.GRU semantics, score.py:83-89,116 /
condition.py:173-179,212) against each other and the epoch-tag machinery of the ring kernel."""
import ctypes

import pytest
import torch

import restatement as O
from helpers import experiments_built, get_spec, record, synth_mix
from test_gpu_parity import get_model, noise_list, run_enhance

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not experiments_built(), reason="round 1's polling-wave kernel is in `make EXPERIMENTS=1` builds only")
@pytest.mark.parametrize("name,B", [("PP16", 1), ("PP16", 2), ("PP24", 1), ("PP24s", 2), ("PP16m", 9)])
def test_ring_kernel_matches_polling_wave_kernel(name, B, steer):
    """option gru_v = 1: first-generation kernel (one polling wave, LDS hand-over, memset per launch); option gru_v = 2: ring kernel
    (every wave gathers from L2, epoch tags; the default at batch 1).  Same recurrence, different summation order
    inside a row.  B = 9 covers batches that are not a multiple of the 8 XCDs, PP24 the 24-workgroup clusters."""
    model, spec, sd = get_model(name)
    T = spec.tot_ds * 37 + 11
    mix = synth_mix(spec, B, T)
    Tp = T + (spec.tot_ds - T % spec.tot_ds)
    nz = noise_list(29, 3, B, Tp)
    steer.set(gru_v=1)
    model.reset_workspace()  # the two generations lay the exchange area out differently: fresh (cleared) workspace each
    ref = run_enhance(model, mix, nz, n_steps=3)
    steer.set(gru_v=2)
    model.reset_workspace()
    out = run_enhance(model, mix, nz, n_steps=3)
    out2 = run_enhance(model, mix, nz, n_steps=3)
    assert torch.equal(out, out2)  # consecutive launches (advancing epochs) are deterministic
    record(f"gru.ring_vs_v1.{name}.b{B}", O.si_sdr(ref, out), 90)
    model.reset_workspace()


@pytest.mark.parametrize("name,B", [("PP16", 1), ("PP16", 3), ("PP24", 1)])
def test_ring_kernel_gather_layouts_agree(name, B, steer):
    """The ring kernel's gather layouts and cluster splits against each other (option gru_upw: 17 = round-2 layout, every unit's
    16 lanes hold all H columns; 16 / 8 = wide layout -- a lane holds its granule pairs for all units of the wave, partial
    sums folded with v_permlane32/16_swap -- with 16 / 8 units per workgroup; 0 = the launcher's choice).  Same recurrence;
    the summation order inside a row differs."""
    model, spec, sd = get_model(name)
    T = spec.tot_ds * 41 + 5
    mix = synth_mix(spec, B, T)
    Tp = T + (spec.tot_ds - T % spec.tot_ds)
    nz = noise_list(37, 3, B, Tp)
    steer.set(gru_upw=17)
    model.reset_workspace()
    ref = run_enhance(model, mix, nz, n_steps=3)
    # (8 units per workgroup at H = 384 are 48-workgroup clusters: more than an XCD's 32 CUs, not offered by the launcher)
    for code in ("16", "8", "0") if name != "PP24" else ("16", "0"):
        steer.set(gru_upw=float(code))
        out = run_enhance(model, mix, nz, n_steps=3)
        assert torch.equal(out, run_enhance(model, mix, nz, n_steps=3))
        record(f"gru.layout_vs_round2.{name}.b{B}.upw{code}", O.si_sdr(ref, out), 90)
    model.reset_workspace()


def test_ring_kernel_epoch_wrap(steer):
    """Tags are epoch + step with a device-side epoch that only grows; near 2^31 the last block of a launch clears the
    exchange area and restarts.  Poke the stored epochs close to the limit and run across it."""
    steer.set(gru_v=2)
    model, spec, sd = get_model("PP16m")
    model.reset_workspace()
    B, T = 2, spec.tot_ds * 25
    mix = synth_mix(spec, B, T - 3)
    nz = noise_list(31, 3, B, T)
    ref = run_enhance(model, mix, nz, n_steps=3)
    hdr = model._ws[:64].view(torch.int32)
    near = 0x7F000000 - 40
    hdr[2] = near  # conditioner exchange area: {epoch, finished blocks}
    hdr[4] = near  # score-net exchange area
    torch.cuda.synchronize()
    out = run_enhance(model, mix, nz, n_steps=3)
    assert torch.equal(ref, out)
    torch.cuda.synchronize()
    assert 0 <= int(hdr[2]) < 10 ** 6 and 0 <= int(hdr[4]) < 10 ** 6  # wrapped and restarted
    assert int(hdr[3]) == 0 and int(hdr[5]) == 0                      # block counters back at zero
    assert torch.equal(ref, run_enhance(model, mix, nz, n_steps=3))
    model.reset_workspace()


def test_ring_kernel_recovers_a_lost_publish(steer):
    """Fault injection (option gru_dbg = 4): one workgroup of every cluster drops its publishes of step 50.  The safety net --
    every waiting wave repeats its last publish as a system-scope store after 256 poll rounds and keeps publishing that
    way for the rest of the launch -- has to bring the pass to the same result, bit for bit, without a timeout.  A publish
    that was never stored looks like a LATE member to the probe (no kind of load sees it): counted as a recovery, not as a
    lost publish, so the handle keeps the cheap publish form."""
    model, spec, sd = get_model("PP16")
    mix = synth_mix(spec, 1, 32000)
    nz = noise_list(61, 2, 1, 32160)
    ref = run_enhance(model, mix, nz, n_steps=2)
    base = model.gru_exchange_stats()
    steer.set(gru_dbg=4)
    out = run_enhance(model, mix, nz, n_steps=2)
    steer.unset("gru_dbg")
    after = model.gru_exchange_stats()
    assert torch.equal(ref, out)
    assert after["recoveries"] > base["recoveries"] and after["system_scope"] > base["system_scope"]
    assert after["lost"] == base["lost"] and "first_event" in after
    assert not model.gru_agent_scope and model._L.ou_get_gru_publish_mode(model._handle) == 0
    assert torch.equal(run_enhance(model, mix, nz, n_steps=2), ref)
    model.reset_workspace()


def test_publish_mode_switches_for_good_when_a_publish_was_invisible(monkeypatch):
    """Status word 33 = recoveries where the awaited granule was there for a system-scope load / an atomic but not for the
    agent-scope load of the gather: the cheap (plain-store) publish form has failed on this device.  The Python wrapper
    (status copy of every call) and the C layer (ou_check_device_status, for callers of the C ABI) both switch the handle to
    agent-scope publishes for good the first time that word moves.  The word is poked by hand here; results do not change."""
    model, spec, sd = get_model("PP16m")
    mix = synth_mix(spec, 1, 4000)
    T = 4000 + (spec.tot_ds - 4000 % spec.tot_ds)
    nz = noise_list(63, 2, 1, T)
    ref = run_enhance(model, mix, nz, n_steps=2)
    assert not model.gru_agent_scope
    hdr = model._ws[:256].view(torch.int32)
    hdr[20] = 3
    hdr[33] = 1
    torch.cuda.synchronize()
    # the C layer on its own
    assert model._L.ou_get_gru_publish_mode(model._handle) == 0
    assert model._L.ou_check_device_status(model._handle, ctypes.c_void_p(model._ws.data_ptr())) == 0
    assert model._L.ou_get_gru_publish_mode(model._handle) == 1
    model._L.ou_set_gru_publish_mode(model._handle, 0)
    # the Python wrapper
    with pytest.warns(RuntimeWarning, match="agent-scope stores from now on"):
        out = run_enhance(model, mix, nz, n_steps=2)
    assert model.gru_agent_scope and model.gru_exchange_stats()["agent_scope_publishes"]
    assert model._L.ou_get_gru_publish_mode(model._handle) == 1
    assert torch.equal(out, ref)
    assert torch.equal(run_enhance(model, mix, nz, n_steps=2), ref)  # same result with write-through publishes
    model._L.ou_set_gru_publish_mode(model._handle, 0)  # (the model is shared by the other tests: back to the default)
    model.gru_agent_scope = False
    model.reset_workspace()
    assert torch.equal(run_enhance(model, mix, nz, n_steps=2), ref)  # and on the fast path again
