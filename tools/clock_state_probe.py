"""Does the +-1.5 % wander of the batch-1 figure follow a device clock domain?  Per-20-call times beside samples of every
pp_dpm_* / hwmon file of THIS GPU's sysfs card (sclk, mclk, fclk, socclk, power, temperature)."""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import open_universe_amd  # noqa: E402,F401
from open_universe_amd import UniverseGAN, config as C, state_dict as S  # noqa: E402

pr = torch.cuda.get_device_properties(0)
want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
card = None
for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
    if os.path.basename(os.path.realpath(c)).lower().startswith(want):
        card = c
print("card", card, want)
files = {}
if card:
    for n in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "gpu_busy_percent", "mem_busy_percent"):
        p = os.path.join(card, n)
        if os.path.exists(p):
            files[n] = p
    for p in glob.glob(os.path.join(card, "hwmon/hwmon*/*_input")) + glob.glob(os.path.join(card, "hwmon/hwmon*/power1_average")):
        files[os.path.basename(p)] = p
print("files", sorted(files))


def read(p):
    try:
        t = open(p).read().strip()
    except OSError:
        return "?"
    if "\n" in t or "*" in t:  # pp_dpm_*: the active level carries a star
        act = [ln for ln in t.splitlines() if "*" in ln]
        return act[0].split(":")[1].strip().rstrip("*").strip() if act else t.replace("\n", "|")
    return t


samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        samples.append((time.perf_counter(), {k: read(p) for k, p in files.items()}))
        time.sleep(0.1)


spec = C.spec_from_config(C.builtin_config("PP16"))
model = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, seed=0), device="cuda:0")
mix = torch.randn(1, 1, 64000, device="cuda:0") * 0.1
rng = torch.Generator(device="cuda:0").manual_seed(1)
for _ in range(5):
    model.enhance(mix, rng=rng)
th = threading.Thread(target=sampler, daemon=True)
th.start()
series = []
for _ in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        model.enhance(mix, rng=rng)
    torch.cuda.synchronize()
    series.append((t0, (time.perf_counter() - t0) / 20 * 1e3))
stop.set()
th.join()
j = 0
last = None
for t0, ms in series:
    while j + 1 < len(samples) and samples[j + 1][0] <= t0:
        j += 1
    cur = samples[j][1] if samples else {}
    diff = {k: v for k, v in cur.items() if last is None or last.get(k) != v}
    print(f"{ms:6.3f} ms  " + " ".join(f"{k}={v}" for k, v in sorted(diff.items())))
    last = cur
