"""Print the figures the docs quote from an evidence run's collected files (profiles/<tag>_*)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06_final"
P = "profiles/" + tag + "_"


def j(name):
    try:
        return json.load(open(P + name))
    except OSError:
        return None


d = j("bench_default.json")
if d:
    r = d["roofline"]
    print(f"headline {d['ms_per_step']:.3f} ms, RTF {d['value']:.0f}, {d['utterances_per_s']:.1f} utt/s; free-running "
          f"{d['free_running'].get('ms_per_enhance', d['free_running'])}")
    print(f"  dominant frac {r['frac']:.3f} achieved {r['achieved']:.1f} executed {r['executed']} avg launch {r['avg_launch_us']:.2f} us "
          f"traffic {r['traffic']}")
    print(f"  gru {r['gru']}")
    print(f"  pointwise {r['pointwise_family']}")
    print(f"  score_forward {r['score_forward']}")
    print(f"  all convs {r['all_conv_kernels']}")
    print(f"  sustained {d['sustained']}")
    print(f"  cpu {d['cpu_baseline']}")
    for k, v in d["in_flight"].items():
        if isinstance(v, dict):
            print(f"  in_flight {k}: {v}")
    print(f"  batch_sweep {d['batch_sweep']}")
for n in ("bench_PP16_b4.json", "bench_PP16_b8.json", "bench_PP16_b16.json", "bench_PP16_b32.json", "bench_C3_PP16_n64_b4.json",
          "bench_C4_OR16_n32_b16.json", "bench_C5_PP24_varlen_b8.json", "bench_force_nccl.json"):
    d = j(n)
    if d:
        r = d["roofline"]
        print(f"{n}: {d['ms_per_step']:.2f} ms, RTF {d['value']:.0f}, {d['utterances_per_s']:.1f} utt/s, dominant {r['kernel'][:40]} "
              f"{r['achieved']:.1f} ({r['frac']:.3f}), all convs {r['all_conv_kernels'].get('frac')}")
for n in ("box_health.txt", "sharded_rate.txt", "lanes_rate.txt", "stress_two_ranks.txt"):
    try:
        print("----", n)
        print(open(P + n).read()[-1500:])
    except OSError:
        pass
