#!/bin/bash
# Round-end evidence run (on the GPU box, from the repo root): bench lines (headline + the other BASELINE configs), rocprofv3
# kernel stats of the headline command, PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) for the conv kernels' memory-side
# traffic of the headline AND the throughput configurations, SQ counters of one deep-level layer (split-K kernel) and one
# throughput layer (no-split-K kernel) in isolation, GRU per-step cycle stamps, the micro-benchmarks behind the design
# decisions, the two-ranks-on-one-GPU stress loop.  Everything lands under gpurun_out/final/; tools/collect_profiles.py copies
# what should be judged into profiles/.
set -u
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 120 python tools/box_health.py 2>&1 | grep "box health" | tee $O/box_health.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
# the same loop under an initialised RCCL group of ONE rank (communicator set-up, device-side weight broadcast, `rccl` block)
timeout 300 python bench.py --gpus 1 --force-nccl --sustained-s 0 --in-flight "" --batch-sweep "" --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_force_nccl.json 2>> $O/bench_default.err
python - $O/bench_force_nccl.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("force-nccl:", d["ms_per_step"], {k: d["rccl"][k] for k in ("ranks", "distinct_device_uuids", "backend", "rccl_version")}, d["weight_broadcast"]["backend"])
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- \
  python bench.py --sustained-s 0 --in-flight "" --steps 10 --warmup 2 --no-cpu-baseline --batch-sweep "" > $O/bench_under_rocprof.json 2> $O/rocprof.err
python tools/kstats.py $O/prof/bench_kernel_trace.csv | head -16 | tee $O/kstats_PP16_B1.txt
# the same command as ONE serial chain (option no_overlap = 1: no side streams inside the call) -- every kernel alone on the device;
# the default run above times the first score-encoder pass beside the conditioner, like bench.py's own per-launch pass does
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o bench -- \
  python bench.py --option no_overlap=1 --sustained-s 0 --in-flight "" --steps 10 --warmup 2 --no-cpu-baseline --batch-sweep "" > $O/bench_under_rocprof_serial.json 2>> $O/rocprof.err
python tools/kstats.py $O/prof_serial/bench_kernel_trace.csv | head -16 | tee $O/kstats_PP16_B1_serial.txt
rm -rf $O/prof_serial
# the other BASELINE configurations (per-GPU shapes): C3 PP16 64 steps B=4, C4 OR16 32 steps B=16, C5 PP24 varlen B=8
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 4 --n_steps 64 --steps 5 --warmup 1 --batch-sweep "" > $O/bench_C3_PP16_n64_b4.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --model OR16 --batch 16 --n_steps 32 --steps 5 --warmup 1 --batch-sweep "" > $O/bench_C4_OR16_n32_b16.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --model PP24 --batch 8 --varlen --steps 5 --warmup 1 > $O/bench_C5_PP24_varlen_b8.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --batch-sweep "" > $O/bench_PP16_b8.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 4 --steps 10 --warmup 2 --no-cpu-baseline --batch-sweep "" > $O/bench_PP16_b4.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 16 --steps 6 --warmup 1 --no-cpu-baseline --batch-sweep "" > $O/bench_PP16_b16.json 2>> $O/bench_default.err
timeout 900 python bench.py --sustained-s 0 --in-flight "" --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --batch-sweep "" > $O/bench_PP16_b32.json 2>> $O/bench_default.err
for f in C3_PP16_n64_b4 C4_OR16_n32_b16 C5_PP24_varlen_b8 PP16_b8 PP16_b4 PP16_b16 PP16_b32; do python - $O/bench_$f.json $f <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "%.2f ms per step, %.1f utt/s, dominant %.1f %% of peak, all conv %.1f %%" % (d["ms_per_step"], d["utterances_per_s"], 100 * d["roofline"]["frac"], 100 * d["roofline"]["all_conv_kernels"]["frac"]))
PY
done
for cfgname in "PP16_B8 --batch 8 --steps 2 --warmup 1" "PP16_B16 --batch 16 --steps 2 --warmup 1" "C3 --batch 4 --n_steps 64 --steps 2 --warmup 1" "C4 --model OR16 --batch 16 --n_steps 32 --steps 2 --warmup 1" "C5 --model PP24 --batch 8 --varlen --steps 2 --warmup 1"; do
  set -- $cfgname; name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$name -o k -- python bench.py --sustained-s 0 --in-flight "" "$@" --no-cpu-baseline --profile-steps 1 --batch-sweep "" > /dev/null 2>> $O/rocprof.err
  python tools/kstats.py $O/prof_$name/k_kernel_trace.csv | head -12 > $O/kstats_$name.txt
  rm -rf $O/prof_$name
done
# memory-side traffic per kernel family: headline, batch 8, C5
for cfgname in "PP16_b1 " "PP16_b8 --batch 8" "PP16_b16 --batch 16" "PP24_b8_varlen --model PP24 --batch 8 --varlen"; do
  set -- $cfgname; tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${tag}_$c -o p -- \
      python bench.py --sustained-s 0 --in-flight "" "$@" --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 --batch-sweep "" > /dev/null 2> $O/pmc_${tag}_$c.err
    # per-family averages now, the raw per-dispatch CSV (tens of MB per pass) stays on the box: gpurun returns 64 MiB at most
    python tools/collect_profiles.py --reduce $O/pmc_${tag}_$c/p_counter_collection.csv $O/pmc_${tag}_$c.json && rm -rf $O/pmc_${tag}_$c
  done
done
# SQ counters (4 per pass): the 512-channel k3 latent-level conv at batch 1 (split-K kernel) and the 192-channel k3 conv of
# UNIVERSE++ 24 kHz at batch 8 (no-split-K throughput kernel), each in isolation
rm -f $O/pmc_sq_latent_conv.txt $O/pmc_sq_direct3_PP24_b8.txt
for set in "SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq -o s -- python tools/conv_one.py _edm_model.encoder.ds_modules.4.conv2 401 20 > /dev/null 2>> $O/rocprof.err
  python tools/pmc_summary.py $O/sq/s_counter_collection.csv conv_direct >> $O/pmc_sq_latent_conv.txt
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq3 -o s -- python tools/conv_one.py _edm_model.encoder.ds_modules.2.conv2 16040 10 PP24 8 > $O/sq3_run.txt 2>> $O/rocprof.err
  python tools/pmc_summary.py $O/sq3/s_counter_collection.csv conv_direct3 >> $O/pmc_sq_direct3_PP24_b8.txt
done
cat $O/sq3_run.txt >> $O/pmc_sq_direct3_PP24_b8.txt
cat $O/pmc_sq_latent_conv.txt $O/pmc_sq_direct3_PP24_b8.txt
rm -rf $O/sq $O/sq3
timeout 300 python tools/gru_ts.py 2>&1 | grep -v amdgpu.ids > $O/gru_ts.txt; tail -5 $O/gru_ts.txt
OU_GRU_AGENT_STORES=1 timeout 300 python tools/gru_ts.py 2>&1 | grep -v amdgpu.ids | sed "s/^/agent-scope publishes: /" >> $O/gru_ts.txt
timeout 300 python tools/direct_ts.py 2>&1 | grep -v amdgpu.ids > $O/direct_ts.txt; tail -8 $O/direct_ts.txt
for args in "PP24 8" "PP16 8" "PP16 4" "OR16 16"; do set -- $args; timeout 300 python tools/tile_sweep.py $1 $2 2>&1 | grep -v amdgpu.ids > $O/tile_sweep_$1_B$2.txt; done
timeout 300 python tools/sharded_rate.py PP16 32 2>&1 | grep -v amdgpu.ids > $O/sharded_rate.txt; cat $O/sharded_rate.txt
# round 4: the wide-load 1x1 / rate-change kernel against the first generation, per layer and tile shape (the EXPERIMENTS library
# has the 64-row shapes too); its phase stamps; ragged sets through K lanes; what runs beside what; batch x lanes; free-running
for b in 1 4 8; do OU_LIBRARY=$PWD/open-universe_amd/lib/libouniverse_experiments.so timeout 600 python tools/d4_sweep.py PP16 $b 2>&1 | grep -v amdgpu.ids | cut -c1-330 > $O/d4_sweep_PP16_B$b.txt; tail -1 $O/d4_sweep_PP16_B$b.txt; done
OU_LIBRARY=$PWD/open-universe_amd/lib/libouniverse_experiments.so timeout 300 python tools/d4_ts.py 1 2>&1 | grep -v amdgpu.ids > $O/d4_ts_B1.txt
timeout 900 python tools/lanes_rate.py PP16 32 1,2,3,4,6,8 2>&1 | grep -v amdgpu.ids | tee $O/lanes_rate.txt
for K in 1 2 4; do timeout 300 python tools/lanes_timeline.py $K 32 2>&1 | grep -v amdgpu.ids; done > $O/lanes_timeline.txt
timeout 600 python tools/lanes_batch.py 64 2>&1 | grep -v amdgpu.ids > $O/lanes_batch.txt
{ timeout 300 python tools/free_run.py; OU_NO_OVERLAP=1 timeout 300 python tools/free_run.py; } 2>&1 | grep -v amdgpu.ids > $O/free_run.txt
# round 5: the wide-load split-K family per layer (8 / 4 slices, minimal filtering, 16-row tiles, XCD ownerships), phase stamps of
# the minimal-filtering fused ConvBlock bodies, the two open questions of round 4 (GPU_MAX_HW_QUEUES=2, three lanes)
timeout 600 python tools/d2_sweep.py PP16 1 -1,2,3 2>&1 | grep -v amdgpu.ids | cut -c1-420 > $O/d2_sweep_PP16_B1.txt; tail -3 $O/d2_sweep_PP16_B1.txt | cut -c1-200
{ timeout 300 python tools/chainw_ts.py score.enc0; timeout 300 python tools/chainw_ts.py score.dec4; timeout 300 python tools/chainw_ts.py score.enc1; } 2>&1 | grep -v amdgpu.ids | cut -c1-330 > $O/chainw_ts.txt; cat $O/chainw_ts.txt
bash tools/hwq_probe.sh $O/hwq > /dev/null 2>&1; cp $O/hwq/hwq_probe.txt $O/hwq_probe.txt; cp $O/hwq/lanes3_probe.txt $O/lanes_three_probe.txt; rm -rf $O/hwq; cat $O/hwq_probe.txt | cut -c1-200
# microbenchmarks behind the design decisions
for u in xchg_latency launch_overhead vmem_issue; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o /tmp/$u 2>> $O/rocprof.err && timeout 150 /tmp/$u > $O/ubench_$u.txt 2>&1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/xcc_migrate.hip -o /tmp/xcc_migrate 2>> $O/rocprof.err
echo "--- one process" > $O/xcc_migrate.txt; timeout 60 /tmp/xcc_migrate 200 2 >> $O/xcc_migrate.txt
echo "--- three processes at once" >> $O/xcc_migrate.txt
for i in 1 2 3; do timeout 120 /tmp/xcc_migrate 300 6 >> $O/xcc_migrate.txt 2>&1 & done; wait
# two ranks on ONE GPU, repeated (the arrangement that produced the GRU time-outs of round 2)
ok=0; bad=0
for i in $(seq 1 30); do
  if timeout 120 python bench.py --sustained-s 0 --in-flight "" --gpus 2 --share-devices --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 1 --batch-sweep "" > $O/st.out 2> $O/st.err; then ok=$((ok+1)); else bad=$((bad+1)); grep -h RuntimeError $O/st.err | head -2 | cut -c1-700 >> $O/stress_two_ranks.txt; fi
done
echo "two ranks on one GPU, bench.py --gpus 2 --share-devices --steps 3: ok=$ok failed=$bad of 30" | tee -a $O/stress_two_ranks.txt
python - >> $O/stress_two_ranks.txt <<PY
import json
d=json.loads([l for l in open("$O/st.out") if l.startswith("{")][-1]); print("last run:", d["ms_per_step"], d["gru_exchange"])
PY
# per-layer tables: the library's own per-launch records paired with its OU_TRACE lines in launch order (tools/layer_table.py)
timeout 600 python tools/layer_table.py PP16 1 2>&1 | grep -v amdgpu.ids > $O/layers_PP16_B1.txt; tail -2 $O/layers_PP16_B1.txt
OU_NO_OVERLAP=1 timeout 600 python tools/layer_table.py PP16 1 2>&1 | grep -v amdgpu.ids > $O/layers_PP16_B1_serial.txt
timeout 600 python tools/layer_table.py PP16 8 2>&1 | grep -v amdgpu.ids > $O/layers_PP16_B8.txt
timeout 600 python tools/layer_table.py PP24 8 2>&1 | grep -v amdgpu.ids > $O/layers_PP24_B8.txt
timeout 600 python tools/layer_table.py PP16 16 2>&1 | grep -v amdgpu.ids > $O/layers_PP16_B16.txt
OU_SPLIT=0 timeout 600 python tools/layer_table.py PP16 16 2>&1 | grep -v amdgpu.ids > $O/layers_PP16_B16_fp32_only.txt
# round 5, late: conv_split_kernel (BF16 matrix pipe, three bf16 pieces per fp32 operand) -- the layer shapes of the table in DESIGN.md
# 4.1f in the microbenchmark (numerics against a double evaluation, phase stamps), the card's clock / power under it, A / B of the product
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -o split_conv.bin split_conv.hip 2>> ../../$O/rocprof.err)
{ for args in "256 256 5 2005 16" "256 256 3 2005 16" "512 512 5 401 16" "512 512 3 401 16" "256 256 5 2005 8" "256 256 3 2005 8" "128 128 5 8020 8 915" "128 128 3 8020 8 913" "128 128 5 8020 16" "64 64 5 32080 8" "64 64 3 32080 8" "512 512 5 401 8" "512 512 3 401 8" "768 768 5 601 8" "256 256 5 2003 3" "64 64 3 301 2"; do
    OU_TS=1 timeout 120 tools/ubench/split_conv.bin $args 200; done; } 2>&1 | grep -v amdgpu.ids > $O/split_ubench.txt; tail -4 $O/split_ubench.txt
timeout 600 python tools/split_clock.py 4 2>&1 | grep -v amdgpu.ids > $O/split_clock.txt; cat $O/split_clock.txt
Q="--sustained-s 0 --in-flight= --no-cpu-baseline --batch-sweep= --profile-steps 0"
for s in 0 -1 0 -1; do
  for cfg in "--batch 16 --steps 5 --warmup 1" "--batch 32 --steps 3 --warmup 1" "--model OR16 --batch 16 --n_steps 32 --steps 3 --warmup 1"; do
    echo "OU_SPLIT=$s $cfg: $(timeout 600 python bench.py --option split=$s $Q $cfg 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("%.2f ms per step, %.1f utt/s" % (d["ms_per_step"], d["utterances_per_s"]))')"
  done
done > $O/split_ab.txt 2>&1; cat $O/split_ab.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/cumask_probe tools/ubench/cumask_probe.hip 2>> $O/rocprof.err && timeout 120 /tmp/cumask_probe > $O/cumask_probe.txt 2>&1
# what travels back (gpurun merges at most 64 MiB): the kernel trace / stats of the headline run, no other rocprofv3 directories
find $O -mindepth 1 -maxdepth 1 -type d ! -name prof -exec rm -rf {} +
find $O/prof -type f ! -name "bench_kernel_stats.csv" ! -name "bench_kernel_trace.csv" -delete 2>/dev/null
du -sh $O | tail -1
