#!/usr/bin/env python
"""
bench.py -- headline benchmark of the `model.enhance` hot path on MI355X.

Metric (BASELINE.json): real-time factor (+ utterances/s) of UNIVERSE++ 16 kHz, 8-step enhance.
  step      = one `enhance` call over one batch of synthetic 4 s utterances (default batch 1 per GPU = configs[1])
  value     = whole-job audio seconds enhanced per wall second (inputs already resident in HBM)
  roofline  = the generic conv kernel (conv_mfma_kernel, the dominant kernel): algorithmic FLOPs of its launches
              / their HIP-event durations, against the fp32 MFMA peak (157.3 TFLOP/s) -- plus the HBM view
  cpu_baseline = the CPU oracle (plain-PyTorch restatement of the reference path) on this box's host cores

Usage: python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)


def synth_mix(fs, B, T, seed0):
    """x_i = 0.1 sin(2 pi f_i t)(0.5 + 0.5 sin(2 pi 3 t)) + 0.03 randn, f_i = 110 (1 + i mod 8)."""
    import math

    t = torch.arange(T) / fs
    out = []
    for i in range(B):
        g = torch.Generator().manual_seed(seed0 + i)
        f = 110.0 * (1 + (seed0 + i) % 8)
        out.append(0.1 * torch.sin(2 * math.pi * f * t) * (0.5 + 0.5 * torch.sin(2 * math.pi * 3 * t))
                   + 0.03 * torch.randn(T, generator=g))
    return torch.stack(out)


def cpu_baseline_worker(model_name, n_steps, seconds, budget_s):
    """Runs in a child process: the oracle timed on the host cores (torch's default intra-op thread count)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restatement as O
    import open_universe_amd  # noqa: F401
    from open_universe_amd import config as C
    from open_universe_amd import state_dict as S

    spec = C.spec_from_config(C.builtin_config(model_name))
    sd = S.synthetic_state_dict(spec, seed=0)
    cores = torch.get_num_threads()
    T = int(seconds * spec.fs)
    mix = synth_mix(spec.fs, 1, T, 1000)
    sdict = spec.to_dict()
    g = torch.Generator().manual_seed(1028282)
    t0 = time.time()
    O.enhance(sd, sdict, mix, n_steps=n_steps, rng=g)  # warm-up
    warm = time.time() - t0
    times = []
    while len(times) < 3 and (sum(times) + warm) < budget_s:
        t0 = time.time()
        O.enhance(sd, sdict, mix, n_steps=n_steps, rng=g)
        times.append(time.time() - t0)
    if not times:
        times = [warm]
    times.sort()
    med = times[len(times) // 2]
    print("CPU_BASELINE_JSON " + json.dumps({
        "value": seconds / med,
        "unit": "x_realtime",
        "utterances_per_s": 1.0 / med,
        "cores": cores,
        "kind": "port",
        "sample": f"1 utterance of {seconds:.0f} s, {n_steps} steps, median of {len(times)} run(s) after 1 warm-up "
                  f"({med:.2f} s per enhance, {cores} torch threads of {os.cpu_count()} logical CPUs); "
                  "oracle/restatement.py (validated against the imported reference)",
    }))


def cpu_baseline(model_name, n_steps, seconds, budget_s=25.0, hard_limit_s=150.0):
    """Bounded CPU sample in a child process, so that a pathological host can never stall the GPU bench."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--model", model_name,
           "--n_steps", str(n_steps), "--seconds", str(seconds), "--cpu-budget", str(budget_s)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=hard_limit_s, env=env)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE_JSON "):
                return json.loads(line[len("CPU_BASELINE_JSON "):])
        return {"value": None, "unit": "x_realtime", "cores": None, "kind": "port",
                "sample": "cpu baseline failed: " + (r.stderr or "")[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "x_realtime", "cores": None, "kind": "port",
                "sample": f"cpu baseline did not finish within {hard_limit_s:.0f} s (skipped)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step (configs[1]: 1)")
    ap.add_argument("--model", default="PP16", choices=["PP16", "OR16", "PP24"])
    ap.add_argument("--n_steps", type=int, default=8, help="diffusion steps")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=25.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        cpu_baseline_worker(args.model, args.n_steps, args.seconds, args.cpu_budget)
        return

    import open_universe_amd  # noqa: F401
    from open_universe_amd import Universe, UniverseGAN
    from open_universe_amd import config as C
    from open_universe_amd import distributed as D
    from open_universe_amd import state_dict as S

    rank, local_rank, world = D.init()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    spec = C.spec_from_config(C.builtin_config(args.model))
    sd = S.synthetic_state_dict(spec, seed=0) if rank == 0 else None
    blob = D.broadcast_packed_weights(spec, sd, device)  # ONE RCCL broadcast; no collective in the loop
    cls = UniverseGAN if spec.kind == "universe_gan" else Universe
    model = cls(spec, packed_weights=blob, device=device)
    model.check_status = False  # no host sync inside the timed region; status is checked afterwards

    T = int(args.seconds * spec.fs)
    mix = synth_mix(spec.fs, args.batch, T, 1000 + rank * args.batch).to(device)
    rng = torch.Generator(device=device).manual_seed(1028282 + rank)

    def step():
        return model.enhance(mix, n_steps=args.n_steps, rng=rng)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    model.check_status = True
    model._status()
    assert torch.isfinite(out).all()

    # ---- roofline of the dominant kernel: per-launch HIP events (separate profiled pass, same workload) ----
    roofline = None
    if rank == 0:
        model.check_status = False
        model.profile(True)
        for _ in range(max(1, args.profile_steps)):
            step()
        torch.cuda.synchronize()
        recs = model.profile_read(max_records=16384)
        model.profile(False)
        # records: (ms, algorithmic flops, algorithmic bytes, variant); variant < 100: conv_mfma_kernel tile configs,
        # >= 100: conv_chain_kernel (fused ConvBlock body).  The roofline entry is the generic kernel -- the largest
        # share of the enhance among the MFMA kernels; the fused kernel is reported beside it.
        def summarise(rr):
            if not rr:
                return None
            ms_ = sum(r[0] for r in rr)
            fl_ = sum(r[1] for r in rr)
            by_ = sum(r[2] for r in rr)
            return {"launches": len(rr) // max(1, args.profile_steps), "avg_launch_us": 1e3 * ms_ / len(rr),
                    "ms_per_enhance": ms_ / max(1, args.profile_steps),
                    "algorithmic_gflop_per_enhance": fl_ / max(1, args.profile_steps) / 1e9,
                    "algorithmic_GB_per_enhance": by_ / max(1, args.profile_steps) / 1e9,
                    "tflops": fl_ / (ms_ * 1e-3) / 1e12, "gbs": by_ / (ms_ * 1e-3) / 1e9,
                    "algorithmic_bytes_per_launch": by_ / len(rr)}
        gen = summarise([r for r in recs if r[3] < 100])
        fused = summarise([r for r in recs if r[3] >= 100])
        traffic, traffic_note = None, "not collected in this run (PMC passes are separate rocprofv3 runs)"
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and args.model == "PP16" and args.batch == 1:
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("conv_mfma_kernel_bytes_per_launch")
            traffic_note = tj.get("note", "")
        roofline = {
            "kernel": "ou::conv_mfma_kernel (generic fp32-MFMA implicit-GEMM Conv1d, all tile configs)",
            "bound": "mfma",
            "achieved": gen["tflops"],
            "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": gen["tflops"] / FP32_MFMA_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_note": traffic_note,
            "launches": gen["launches"],
            "avg_launch_us": gen["avg_launch_us"],
            "algorithmic_bytes_per_launch": gen["algorithmic_bytes_per_launch"],
            "algorithmic_gflop_per_enhance": gen["algorithmic_gflop_per_enhance"],
            "conv_ms_per_enhance": gen["ms_per_enhance"],
            "hbm_view": {"achieved_GBs": gen["gbs"], "peak_GBs": HBM_PEAK_GBS, "frac": gen["gbs"] / HBM_PEAK_GBS,
                         "algorithmic_GB_per_enhance": gen["algorithmic_GB_per_enhance"]},
            "fused_block_kernel": None if fused is None else {
                "kernel": "ou::conv_chain_kernel (fused ConvBlock body, C = 32 / 64 levels)",
                "achieved": fused["tflops"], "frac": fused["tflops"] / FP32_MFMA_PEAK_TFLOPS,
                "launches": fused["launches"], "avg_launch_us": fused["avg_launch_us"],
                "ms_per_enhance": fused["ms_per_enhance"],
                "algorithmic_gflop_per_enhance": fused["algorithmic_gflop_per_enhance"]},
            "method": f"device-side per-launch timing (first block start .. last block end on the 100 MHz s_memrealtime clock) of every conv launch, profiled pass of {args.profile_steps} "
                      "enhance calls right after the timed region; algorithmic FLOPs/bytes = reference (un-folded) "
                      "layer-granular accounting, SURVEY.md 8(d)",
        }

    if rank == 0:
        audio_s = args.steps * args.batch * args.seconds * world
        res = {
            "metric": "real_time_factor (audio s / wall s), UNIVERSE++ 16 kHz 8-step enhance",
            "value": audio_s / dt,
            "unit": "x_realtime",
            "utterances_per_s": args.steps * args.batch * world / dt,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded AM-sine + noise waveforms; seeded random weights with the reference key schema)",
            "config": {
                "workload": f"UNIVERSE++ 16 kHz, {args.n_steps} diffusion steps, batch={args.batch} utterance(s) of "
                            f"{args.seconds:.0f} s per GPU per step" if args.model == "PP16" else
                            f"{args.model}, {args.n_steps} steps, batch={args.batch}, {args.seconds:.0f} s",
                "model": args.model,
                "n_diffusion_steps": args.n_steps,
                "batch_per_gpu": args.batch,
                "samples_per_utterance": T,
                "parallelism": "utterances sharded across GPUs; packed weights broadcast once over RCCL; "
                               "no collective in the sampling loop",
            },
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.model, args.n_steps, args.seconds)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
