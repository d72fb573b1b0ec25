#!/bin/bash
# Round-end evidence run (on the GPU box, from the repo root): bench lines (headline + the other BASELINE configs),
# rocprofv3 kernel stats of the headline command, the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) for the conv
# kernels' memory-side traffic, SQ counters of one deep-level conv layer in isolation, GRU per-step cycle stamps.
# Everything lands under gpurun_out/final/; tools/collect_profiles.py copies what should be judged into profiles/.
set -u
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- \
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
python tools/kstats.py $O/prof/bench_kernel_trace.csv | head -16 | tee $O/kstats_PP16_B1.txt
# the other BASELINE configurations (per-GPU shapes): C3 PP16 64 steps B=4, C4 OR16 32 steps B=16, C5 PP24 varlen B=8
timeout 900 python bench.py --batch 4 --n_steps 64 --steps 5 --warmup 1 > $O/bench_C3_PP16_n64_b4.json 2>> $O/bench_default.err
timeout 900 python bench.py --model OR16 --batch 16 --n_steps 32 --steps 5 --warmup 1 > $O/bench_C4_OR16_n32_b16.json 2>> $O/bench_default.err
timeout 900 python bench.py --model PP24 --batch 8 --varlen --steps 5 --warmup 1 > $O/bench_C5_PP24_varlen_b8.json 2>> $O/bench_default.err
timeout 900 python bench.py --batch 8 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_PP16_b8.json 2>> $O/bench_default.err
timeout 900 python bench.py --batch 4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_PP16_b4.json 2>> $O/bench_default.err
for f in C3_PP16_n64_b4 C4_OR16_n32_b16 C5_PP24_varlen_b8 PP16_b8 PP16_b4; do tail -c 200 $O/bench_$f.json | head -c 200; echo; done
for cfgname in "PP16_B8 --batch 8 --steps 2 --warmup 1" "C3 --batch 4 --n_steps 64 --steps 2 --warmup 1" "C4 --model OR16 --batch 16 --n_steps 32 --steps 2 --warmup 1" "C5 --model PP24 --batch 8 --varlen --steps 2 --warmup 1"; do
  set -- $cfgname; name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$name -o k -- python bench.py "$@" --no-cpu-baseline --profile-steps 1 > /dev/null 2>> $O/rocprof.err
  python tools/kstats.py $O/prof_$name/k_kernel_trace.csv | head -10 > $O/kstats_$name.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- \
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > /dev/null 2> $O/pmc_$c.err
done
# SQ counters (4 per pass) of the 512-channel k3 latent-level conv in isolation (direct kernel)
L=_edm_model.encoder.ds_modules.4.conv2
rm -f $O/pmc_sq_latent_conv.txt
for set in "SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq -o s -- python tools/conv_one.py $L 401 20 > /dev/null 2>> $O/rocprof.err
  python tools/pmc_summary.py $O/sq/s_counter_collection.csv conv_direct >> $O/pmc_sq_latent_conv.txt
done
cat $O/pmc_sq_latent_conv.txt
timeout 300 python tools/gru_ts.py 2>&1 | grep -v amdgpu.ids > $O/gru_ts.txt; tail -5 $O/gru_ts.txt
timeout 300 python tools/direct_ts.py 2>&1 | grep -v amdgpu.ids > $O/direct_ts.txt; tail -8 $O/direct_ts.txt
timeout 300 python tools/direct_sweep.py PP16 1 2>&1 | grep -v amdgpu.ids > $O/direct_sweep_PP16_B1.txt
# microbenchmarks behind the design decisions (exchange hand-off latency, dispatch cost, VMEM issue rate)
for u in xchg_latency launch_overhead vmem_issue; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o /tmp/$u 2>> $O/rocprof.err && timeout 150 /tmp/$u > $O/ubench_$u.txt 2>&1
done
OU_TRACE=1 OU_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python tools/gpu_debug.py timing PP16 iters=2 > /dev/null 2> $O/trace.log
python tools/trace_summary.py $O/tr/t_kernel_trace.csv $O/trace.log > $O/layers_PP16_B1.txt 2>&1
