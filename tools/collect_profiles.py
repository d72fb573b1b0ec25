"""Copy the round-end evidence from gpurun_out/final/ (tools/round_profile.sh) into profiles/ and derive
profiles/pmc_traffic.json (read by bench.py for roofline.traffic -- labelled STATIC there)."""
import csv, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "final")
dst = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06_final"
FAMS = {"direct2": ("conv_direct2_kernel", "conv_direct2w_kernel", "conv_direct4w_kernel"),
        "direct": ("conv_direct_kernel", "conv_direct_strided_kernel"),
        "direct4": ("conv_direct4_kernel",),
        "direct3": ("conv_direct3_kernel", "conv_direct3w_kernel", "conv_direct3s_kernel"),
        "split": ("conv_split_kernel",),
        "lds": ("conv_mfma_kernel",), "rate": ("rate_down_kernel", "rate_up_kernel"),
        "chain": ("conv_chain_kernel", "conv_chainw_kernel"), "gru_ring": ("gru_ring_kernel",),
        "gru_cluster": ("gru_cluster_kernel",)}


def fam_avg(path):
    if path.endswith(".json"):  # already reduced on the GPU box (tools/round_profile.sh: the raw per-dispatch CSVs of four
        return json.load(open(path))  # configurations x two counters no longer fit gpurun's 64 MiB return limit)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        for fam, pats in FAMS.items():
            if any(pat in r["Kernel_Name"] for pat in pats):
                agg[fam].append(float(r["Counter_Value"]))
    return {k: {"dispatches": len(v), "avg": sum(v) / len(v)} for k, v in agg.items()}


if len(sys.argv) > 3 and sys.argv[1] == "--reduce":
    json.dump(fam_avg(sys.argv[2]), open(sys.argv[3], "w"))
    sys.exit(0)

shutil.copy(os.path.join(src, "prof", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats_PP16_B1.csv"))
for n in sorted(os.listdir(src)):
    if n.startswith("bench_") and n.endswith(".json"):
        lines = [l for l in open(os.path.join(src, n)).read().strip().splitlines() if l.startswith("{")]
        if not lines:
            print("skip (no JSON line):", n)
            continue
        json.loads(lines[-1])
        open(os.path.join(dst, f"{tag}_{n}"), "w").write(lines[-1] + "\n")
for n in sorted(os.listdir(src)):
    if n.startswith(("kstats_", "pmc_sq_", "gru_ts", "layers", "direct_ts", "direct_sweep", "ubench_", "tile_sweep", "stress_",
                     "xcc_migrate", "timings", "sharded_rate", "box_health", "d4_sweep", "d4_ts", "lanes_", "free_run", "d2_sweep", "chainw_ts",
                     "hwq", "env_knobs", "split_", "cumask", "summary")):
        shutil.copy(os.path.join(src, n), os.path.join(dst, f"{tag}_{n}"))

KB = 1024.0
NOTE = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate runs of the bench command of that configuration with "
        "--steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1), averaged per dispatch over the instantiations of each kernel "
        "family; bytes = 2 x FETCH_SIZE KB (gfx950 correction of MI355X_MICROARCH.md, HBM section: the counter tallies 128-B "
        "requests at 64 B) + WRITE_SIZE KB (uncalibrated, taken as is). Memory-side L2 traffic: Infinity-Cache hits are included.")
out = {"note": NOTE, "configs": {}}
# configuration tag -> sub-directory prefix written by tools/round_profile.sh (pmc_<tag>_FETCH_SIZE / _WRITE_SIZE)
for cfg_tag in ("PP16_b1", "PP16_b8", "PP16_b16", "PP24_b8_varlen", "PP16_b4_n64", "OR16_b16_n32"):
    fp = os.path.join(src, f"pmc_{cfg_tag}_FETCH_SIZE", "p_counter_collection.csv")
    wp = os.path.join(src, f"pmc_{cfg_tag}_WRITE_SIZE", "p_counter_collection.csv")
    if not (os.path.exists(fp) and os.path.exists(wp)):
        fp, wp = os.path.join(src, f"pmc_{cfg_tag}_FETCH_SIZE.json"), os.path.join(src, f"pmc_{cfg_tag}_WRITE_SIZE.json")
    if not (os.path.exists(fp) and os.path.exists(wp)):
        continue
    fetch, write = fam_avg(fp), fam_avg(wp)
    ent = {"raw_KB_per_dispatch": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}}
    for fam in FAMS:
        if fam in fetch and fam in write:
            ent[fam + "_bytes_per_launch"] = (2 * fetch[fam]["avg"] + write[fam]["avg"]) * KB
            ent[fam + "_bytes_per_launch_uncorrected"] = (fetch[fam]["avg"] + write[fam]["avg"]) * KB
    out["configs"][cfg_tag] = ent
    if cfg_tag == "PP16_b1":  # (the keys bench.py of round 2 read, kept at the top level)
        out.update({k: v for k, v in ent.items()})
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({c: {k: round(v) for k, v in e.items() if k.endswith("per_launch")} for c, e in out["configs"].items()}, indent=1))
