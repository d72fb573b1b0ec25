"""Shader clock and board power of THIS GPU while conv_split_kernel (BF16 matrix pipe) runs back to back, beside the same layer on
the fp32 kernels: is the 2.5 PFLOP/s pipe's clock the 2.4 GHz the roofline assumes?  (bench.GpuSampler: amdgpu hwmon of the card
that matches the HIP device's PCI address.)
usage: split_clock.py [seconds=4]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
torch.cuda.init()


def leg(name, cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    with bench.GpuSampler(period_s=0.2) as s:
        t0 = time.perf_counter()
        out = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=300).stdout
        dt = time.perf_counter() - t0
    sm = s.summary()
    f, w = sm.get("sclk_MHz", {}), sm.get("power_W", {})
    tail = [l.strip() for l in out.strip().splitlines() if "us" in l][-1:] or [out.strip()[-200:]]
    # (the first samples cover process start-up and the CPU-side check: report the steady tail)
    tr = [x for x in sm["trace_t_s_sclk_MHz_power_W"] if x[0] > dt * 0.6]
    fs = [x[1] for x in tr if x[1]]
    ws = [x[2] for x in tr if x[2]]
    print(f"{name}: {dt:.1f} s; sclk over the last 40 % of the leg {min(fs):.0f} .. {max(fs):.0f} MHz (whole leg {f.get('min')} .. {f.get('max')}), "
          f"power {min(ws):.0f} .. {max(ws):.0f} W (card {sm['card']})" if fs and ws else f"{name}: no samples ({sm['source']})")
    for l in tail:
        print("    " + l)


ub = os.path.join(ROOT, "tools", "ubench", "split_conv.bin")
for shape, us in (("256 256 5 2005 16 825", 118.0), ("128 128 5 8020 8 915", 66.0)):
    iters = int(secs * 1e6 / us)
    leg(f"conv_split_kernel {shape} x {iters}", [ub] + shape.split() + [str(iters)])
# the same 256-channel k5 layer on the fp32 minimal-filtering kernel (conv_direct3w_kernel)
leg("fp32 kernels, encoder.ds_modules.3.conv1 (256 ch, k5) B=16 (OU_SPLIT=0)", [sys.executable, os.path.join(ROOT, "tools", "conv_one.py"), "_edm_model.encoder.ds_modules.3.conv1", "2005", str(int(secs * 1e6 / 131)), "PP16", "16"],
    {"OU_SPLIT": "0"})
leg("bf16-split kernel through the library, encoder.ds_modules.3.conv1 (256 ch, k5) B=16", [sys.executable, os.path.join(ROOT, "tools", "conv_one.py"), "_edm_model.encoder.ds_modules.3.conv1", "2005", str(int(secs * 1e6 / 118)), "PP16", "16"])
