"""Per-layer table of ONE enhance call: every conv / fused-ConvBlock / GRU launch with its own device-side duration.

Pairing is by construction, not by guesswork: the library records every profiled launch (first block start .. last block end on
the device's constant clock, the records bench.py's roofline reads) in HOST LAUNCH ORDER, and prints one OU_TRACE line per conv
launch in the same order -- record i is line i, whatever stream the launch ran on (round 4's table paired a rocprofv3 trace,
sorted by start time, with the OU_TRACE log and mis-assigned the launches of the side streams: cond.st0 at 202 TFLOP/s).
Two rates per layer: ALGORITHMIC (2 M Cin KW Nq B / time: the layer-granular accounting of SURVEY.md 8(d), what roofline.frac
is quoted in) and EXECUTED on the matrix pipe -- the minimal-filtering kernels issue (KW + 1) / (2 KW) of the algorithmic
multiply-adds (2/3 for k3, 3/5 for k5), so their algorithmic rate may exceed the 157.3 TFLOP/s of the pipe; the executed one may
not, and the script asserts that.

  python tools/layer_table.py [PP16|OR16|PP24] [B] [n_steps]      (OU_NO_OVERLAP=1 for one serial chain)"""
import os, re, sys, tempfile
os.environ["OU_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import get_spec, synth_mix
from open_universe_amd import Universe, UniverseGAN, state_dict as S
from open_universe_amd.universe import Universe as _U; _U.steer_from_env = True  # tools only: OU_<OPTION>=v env vars -> ou_set_option

PEAK = 157.3
name = sys.argv[1] if len(sys.argv) > 1 else "PP16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
spec = get_spec(name)
cls = UniverseGAN if spec.kind == "universe_gan" else Universe
model = cls(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
mix = synth_mix(spec, B, 4 * spec.fs).cuda()
model.check_status = False
g = torch.Generator(device="cuda").manual_seed(0)
# the library's stderr (one OU_TRACE line per conv launch) is captured from here on: two warm-up calls, then the profiled one
sys.stderr.flush()
saved = os.dup(2)
tmp = tempfile.TemporaryFile(mode="w+b")
os.dup2(tmp.fileno(), 2)
try:
    for _ in range(2):
        model.enhance(mix, n_steps=n_steps, rng=g)
    torch.cuda.synchronize()
    sys.stderr.flush()
    tmp.seek(0)
    tmp.truncate()
    model.profile(True)
    model.enhance(mix, n_steps=n_steps, rng=g)
    torch.cuda.synchronize()
    recs = model.profile_read(max_records=32768)
    model.profile(False)
finally:
    sys.stderr.flush()
    os.dup2(saved, 2)
    os.close(saved)
tmp.seek(0)
lines = [l.split() for l in tmp.read().decode(errors="replace").splitlines() if l.startswith("OU_TRACE conv") or l.startswith("OU_TRACE chain")]
conv_recs = [r for r in recs if r[3] < 1000]
assert len(conv_recs) == len(lines), (len(conv_recs), len(lines))


def executed_fraction(cfg, kw, depth=None):
    """MFMA multiply-adds issued / algorithmic ones."""
    if 400 <= cfg < 800:
        return (kw + 1) / (2.0 * kw)
    if 800 <= cfg < 1000:  # conv_split_kernel: six bf16 products per MAC on the BF16 pipe (2 500 TFLOP/s dense = 16 x the f32 pipe):
        return 6.0 / 16.0  # in units of the f32 pipe's capacity, so that the one assertion below covers both pipes
    if cfg == 193:   # conv_chainw_kernel, depth 3: k5, k3, k3
        return (6 + 4 + 4) / (10 + 6 + 6.0)
    if cfg == 192:   # depth 2: k3, k3
        return 8 / 12.0
    return 1.0


fam = {}
rows, it = [], iter(lines)
worst = 0.0
for ms, fl, by, cfg in recs:
    us = ms * 1e3
    if cfg >= 1000:
        rows.append(("gru (recurrence, %d steps)" % (cfg - 1000), "", us, fl / (ms * 1e-3) / 1e12, None, cfg))
        key = "gru"
    else:
        l = next(it)
        nm = l[2]
        if l[1] == "chain":
            depth = int(l[4].split("=")[1])
            frac = executed_fraction(cfg, 0)
            shape = " ".join(l[3:])
        else:
            kw = int(re.search(r"KW=(\d+)", " ".join(l)).group(1))
            frac = executed_fraction(cfg, kw)
            shape = " ".join(l[3:9])
        alg = fl / (ms * 1e-3) / 1e12
        rows.append((nm, shape, us, alg, alg * frac, cfg))
        worst = max(worst, alg * frac)
        key = ("direct2/2w/4w" if cfg in (66, 76, 67, 77) or 400 <= cfg < 500 or 600 <= cfg < 800 else "direct (dword)" if 50 <= cfg < 100 else
               "lds" if cfg < 40 else "rate" if cfg < 50 else "chain" if cfg < 200 else "direct3/3w/3s" if cfg < 300 or 500 <= cfg < 600 else "split (bf16 x 6)" if cfg >= 800 else "direct4")
    f = fam.setdefault(key, [0, 0.0, 0.0])
    f[0] += 1; f[1] += us; f[2] += fl
tot_us = sum(r[2] for r in rows)
print(f"{name} B={B} n_steps={n_steps}: {len(rows)} profiled launches (convs, fused ConvBlock bodies, GRU passes), {tot_us/1e3:.3f} ms of device time "
      f"(launches of the side streams overlap: this is NOT the wall time of the call)")
for k, (n, us, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:16s} {n:4d} launches {us/1e3:8.3f} ms  avg {us/n:7.1f} us  {fl/(us*1e-6)/1e12:6.1f} TF/s algorithmic")
print("per launch (conditioner, then the first two score passes; the other passes repeat the second):")
seen = {}
for nm, shape, us, alg, exe, cfg in rows:
    c = seen.get(nm, 0)
    seen[nm] = c + 1
    if c >= (1 if nm.startswith("cond.") else 2):
        continue
    if exe is None:
        print(f"  {nm:34s} {'':62s} {us:8.1f} us {alg:7.1f} TF/s")
    else:
        ex = f" (executed {exe:6.1f})" if abs(exe - alg) > 1e-9 else ""
        if 800 <= cfg < 1000:
            ex = f" (bf16 pipe {alg * 6:6.0f})"
        print(f"  {nm:34s} {shape:62s} {us:8.1f} us {alg:7.1f} TF/s{ex}")
print(f"largest EXECUTED rate of a launch: {worst:.1f} TFLOP/s (fp32 MFMA peak {PEAK})")
assert worst <= PEAK, "a launch above the matrix pipe's peak: the record <-> layer pairing or the FLOP accounting is wrong"
