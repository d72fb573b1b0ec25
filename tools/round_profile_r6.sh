#!/bin/bash
# Round-6 evidence run (on the GPU box, from the repo root): the part of tools/round_profile.sh that this round's changes touch --
# bench lines (headline with the ragged-set leg: serial / lanes / exact batching; the other BASELINE configs; batch 4 .. 32),
# rocprofv3 kernel stats of the headline command (default and serial chain) and of the throughput configs, PMC FETCH / WRITE
# passes of the headline and batch 8, GRU phase stamps, per-layer tables, the two-ranks-on-one-GPU loop.  The kernel sweeps and
# microbenchmarks of rounds 2 - 5 (tile / d2 / d4 sweeps, split ubench, hwq probe ...) are unchanged code: tools/round_profile.sh.
# Everything lands under gpurun_out/final/; `python tools/collect_profiles.py r06_final` copies what is judged into profiles/.
set -u
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O
mkdir -p $O
timeout 120 python tools/box_health.py 2>&1 | grep "box health" | tee $O/box_health.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json; echo
timeout 300 python bench.py --gpus 1 --force-nccl --sustained-s 0 --in-flight "" --batch-sweep "" --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_force_nccl.json 2>> $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- \
  python bench.py --sustained-s 0 --in-flight "" --steps 10 --warmup 2 --no-cpu-baseline --batch-sweep "" > $O/bench_under_rocprof.json 2> $O/rocprof.err
python tools/kstats.py $O/prof/bench_kernel_trace.csv | head -16 | tee $O/kstats_PP16_B1.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o bench -- \
  python bench.py --option no_overlap=1 --sustained-s 0 --in-flight "" --steps 10 --warmup 2 --no-cpu-baseline --batch-sweep "" > $O/bench_under_rocprof_serial.json 2>> $O/rocprof.err
python tools/kstats.py $O/prof_serial/bench_kernel_trace.csv | head -16 | tee $O/kstats_PP16_B1_serial.txt
rm -rf $O/prof_serial
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 4 --n_steps 64 --steps 5 --warmup 1 --batch-sweep "" > $O/bench_C3_PP16_n64_b4.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --model OR16 --batch 16 --n_steps 32 --steps 5 --warmup 1 --batch-sweep "" > $O/bench_C4_OR16_n32_b16.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --model PP24 --batch 8 --varlen --steps 5 --warmup 1 > $O/bench_C5_PP24_varlen_b8.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --batch-sweep "" > $O/bench_PP16_b8.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 4 --steps 10 --warmup 2 --no-cpu-baseline --batch-sweep "" > $O/bench_PP16_b4.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 16 --steps 6 --warmup 1 --no-cpu-baseline --batch-sweep "" > $O/bench_PP16_b16.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --batch-sweep "" > $O/bench_PP16_b32.json 2>> $O/bench_default.err
for f in default C3_PP16_n64_b4 C4_OR16_n32_b16 C5_PP24_varlen_b8 PP16_b8 PP16_b4 PP16_b16 PP16_b32; do python - $O/bench_$f.json $f <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "%.2f ms per step, %.1f utt/s, dominant %.1f %% of peak, all conv %.1f %%" % (d["ms_per_step"], d["utterances_per_s"], 100 * d["roofline"]["frac"], 100 * d["roofline"]["all_conv_kernels"]["frac"]))
PY
done | tee $O/summary.txt
for cfgname in "PP16_B8 --batch 8 --steps 2 --warmup 1" "PP16_B16 --batch 16 --steps 2 --warmup 1" "C3 --batch 4 --n_steps 64 --steps 2 --warmup 1" "C4 --model OR16 --batch 16 --n_steps 32 --steps 2 --warmup 1" "C5 --model PP24 --batch 8 --varlen --steps 2 --warmup 1"; do
  set -- $cfgname; name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$name -o k -- python bench.py --sustained-s 0 --in-flight "" "$@" --no-cpu-baseline --profile-steps 1 --batch-sweep "" > /dev/null 2>> $O/rocprof.err
  python tools/kstats.py $O/prof_$name/k_kernel_trace.csv | head -12 > $O/kstats_$name.txt
  rm -rf $O/prof_$name
done
# a ragged batch under rocprofv3: which kernels carry the per-row lengths, and what is left of the separate mask launches
cat > /tmp/ragged_run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
from helpers import get_spec, synth_mix
from open_universe_amd import UniverseGAN, state_dict as S, distributed as D
spec = get_spec("PP16")
m = UniverseGAN(spec, state_dict=S.synthetic_state_dict(spec, 0), device="cuda:0")
g = torch.Generator().manual_seed(17)
lens = sorted({int(spec.fs * 4.0 * (0.875 + 0.125 * float(torch.rand(1, generator=g)))) for _ in range(40)})[:32]
sigs = [synth_mix(spec, 1, n, 2000 + i)[0].cuda() for i, n in enumerate(lens)]
for _ in range(2):
    D.enhance_sharded(m, sigs, seed=3, gather=False, batch_size=8, n_steps=8)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_ragged -o k -- python /tmp/ragged_run.py > /dev/null 2>> $O/rocprof.err
python tools/kstats.py $O/prof_ragged/k_kernel_trace.csv | head -24 > $O/kstats_ragged_PP16_bs8.txt; grep -i "mask\|tail_fill\|_var" $O/kstats_ragged_PP16_bs8.txt
rm -rf $O/prof_ragged
for cfgname in "PP16_b1 " "PP16_b8 --batch 8"; do
  set -- $cfgname; tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${tag}_$c -o p -- \
      python bench.py --sustained-s 0 --in-flight "" "$@" --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 --batch-sweep "" > /dev/null 2> $O/pmc_${tag}_$c.err
    python tools/collect_profiles.py --reduce $O/pmc_${tag}_$c/p_counter_collection.csv $O/pmc_${tag}_$c.json && rm -rf $O/pmc_${tag}_$c
  done
done
timeout 300 python tools/gru_ts.py 2>&1 | grep -v amdgpu.ids > $O/gru_ts.txt; tail -5 $O/gru_ts.txt | cut -c1-200
timeout 300 python tools/sharded_rate.py PP16 32 2>&1 | grep -v amdgpu.ids > $O/sharded_rate.txt; cat $O/sharded_rate.txt
timeout 900 python tools/lanes_rate.py PP16 32 1,4,8 2>&1 | grep -v amdgpu.ids | tee $O/lanes_rate.txt
ok=0; bad=0
for i in $(seq 1 12); do
  if timeout 120 python bench.py --sustained-s 0 --in-flight "" --gpus 2 --share-devices --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 1 --batch-sweep "" > $O/st.out 2> $O/st.err; then ok=$((ok+1)); else bad=$((bad+1)); grep -h RuntimeError $O/st.err | head -2 | cut -c1-700 >> $O/stress_two_ranks.txt; fi
done
echo "two ranks on one GPU, bench.py --gpus 2 --share-devices --steps 3: ok=$ok failed=$bad of 12" | tee -a $O/stress_two_ranks.txt
python - >> $O/stress_two_ranks.txt <<PY
import json
d=json.loads([l for l in open("$O/st.out") if l.startswith("{")][-1]); print("last run:", d["ms_per_step"], d["utterances_per_s_per_gpu"], d["rank0_alone"], d["gru_exchange"])
PY
timeout 600 python tools/layer_table.py PP16 1 2>&1 | grep -v amdgpu.ids > $O/layers_PP16_B1.txt; tail -2 $O/layers_PP16_B1.txt
OU_NO_OVERLAP=1 timeout 600 python tools/layer_table.py PP16 1 2>&1 | grep -v amdgpu.ids > $O/layers_PP16_B1_serial.txt
timeout 600 python tools/layer_table.py PP16 8 2>&1 | grep -v amdgpu.ids > $O/layers_PP16_B8.txt
timeout 600 python tools/layer_table.py PP16 16 2>&1 | grep -v amdgpu.ids > $O/layers_PP16_B16.txt
find $O -mindepth 1 -maxdepth 1 -type d ! -name prof -exec rm -rf {} +
find $O/prof -type f ! -name "bench_kernel_stats.csv" ! -name "bench_kernel_trace.csv" -delete 2>/dev/null
du -sh $O | tail -1
