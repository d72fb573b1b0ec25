"""Copy the round-end evidence from gpurun_out/final/ (tools/round_profile.sh) into profiles/ and derive
profiles/pmc_traffic.json (read by bench.py for roofline.traffic)."""
import csv, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "final")
dst = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
shutil.copy(os.path.join(src, "prof", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats_PP16_B1.csv"))
for n in ("bench_default.json", "bench_under_rocprof.json"):
    line = open(os.path.join(src, n)).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(dst, f"{tag}_{n}"), "w").write(line + "\n")


def fam_avg(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        fam = "conv_mfma_kernel" if "conv_mfma_kernel" in k else "conv_chain_kernel" if "conv_chain_kernel" in k else \
            "gru_cluster_kernel" if "gru_cluster_kernel" in k else "fir_kernel" if "fir_kernel" in k else None
        if fam:
            agg[fam].append(float(r["Counter_Value"]))
    return {k: {"dispatches": len(v), "avg": sum(v) / len(v)} for k, v in agg.items()}


fetch = fam_avg(os.path.join(src, "pmc_FETCH_SIZE", "p_counter_collection.csv"))
write = fam_avg(os.path.join(src, "pmc_WRITE_SIZE", "p_counter_collection.csv"))
KB = 1024.0
out = {
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate runs of `python bench.py --steps 2 --warmup 1 "
            "--no-cpu-baseline --profile-steps 1`, PP16, batch 1), averaged per dispatch over all conv_mfma_kernel "
            "instantiations; bytes = 2 x FETCH_SIZE KB (gfx950 correction of MI355X_MICROARCH.md, HBM section: the counter "
            "tallies 128-B requests at 64 B) + WRITE_SIZE KB (uncalibrated, taken as is). Memory-side L2 traffic: "
            "Infinity-Cache hits are included.",
    "raw_KB_per_dispatch": {"FETCH_SIZE": fetch, "WRITE_SIZE": write},
    "conv_mfma_kernel_bytes_per_launch": (2 * fetch["conv_mfma_kernel"]["avg"] + write["conv_mfma_kernel"]["avg"]) * KB,
    "conv_mfma_kernel_bytes_per_launch_uncorrected": (fetch["conv_mfma_kernel"]["avg"] + write["conv_mfma_kernel"]["avg"]) * KB,
    "conv_chain_kernel_bytes_per_launch": (2 * fetch["conv_chain_kernel"]["avg"] + write["conv_chain_kernel"]["avg"]) * KB
    if "conv_chain_kernel" in fetch else None,
}
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
