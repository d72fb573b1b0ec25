"""Throughput of ragged utterance sets through the product path (`distributed.enhance_sharded`) on one GPU: the serial
one-call-at-a-time loop against K calls in flight (`in_flight=K`, open_universe_amd/lanes.py), and the batched mode on the
equal-length set for reference.   python tools/lanes_rate.py [PP16] [n_utt] [lanes,lanes,..]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from helpers import get_spec, synth_mix  # noqa: E402
from open_universe_amd import Universe, UniverseGAN, distributed as D, state_dict as S  # noqa: E402
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

name = sys.argv[1] if len(sys.argv) > 1 else "PP16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lanes_list = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3,4,6,8").split(",")]
spec = get_spec(name)
cls = UniverseGAN if spec.kind == "universe_gan" else Universe
model = cls(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
fs = spec.fs


def lengths(lo_s, hi_s):
    g = torch.Generator().manual_seed(17)
    out = set()
    while len(out) < n:
        out.add(int(fs * (lo_s + (hi_s - lo_s) * float(torch.rand(1, generator=g)))))
    lens = sorted(out)
    order = torch.randperm(n, generator=g).tolist()
    return [lens[i] for i in order]


sets = {"equal 4.0 s": [int(4 * fs)] * n, "ragged 3.5-4.0 s": lengths(3.5, 4.0), "ragged 1.0-4.0 s": lengths(1.0, 4.0)}
print(f"{name}, {n} utterances per set, 8 steps; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}", flush=True)
for tag, lens in sets.items():
    sigs = [synth_mix(spec, 1, L, seed=400 + i)[0].cuda() for i, L in enumerate(lens)]
    audio_s = sum(lens) / fs
    ref = None
    for k in lanes_list:
        for rep in range(2):  # the first pass also creates the lanes / workspaces
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = D.enhance_sharded(model, sigs, seed=3, gather=False, in_flight=k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        same = ""
        if ref is None:
            ref = outs
        else:
            same = "  bit-identical to lanes=1" if all(torch.equal(ref[i], outs[i]) for i in ref) else "  DIFFERS from lanes=1"
        st = model.gru_exchange_stats()
        print(f"  {tag:18s} in_flight={k}: {1e3 * dt / n:6.2f} ms per utterance, {n / dt:6.1f} utt/s, RTF {audio_s / dt:6.0f}x"
              f"{same}  (recoveries {st['recoveries']})", flush=True)
    if tag.startswith("equal"):
        for bs in (4, 8):
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                D.enhance_sharded(model, sigs, seed=3, gather=False, batch_size=bs)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            print(f"  {tag:18s} batch_size={bs}: {1e3 * dt / n:6.2f} ms per utterance, {n / dt:6.1f} utt/s (batched: equal lengths only)", flush=True)
