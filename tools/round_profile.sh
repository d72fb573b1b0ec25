#!/bin/bash
# Round-end evidence run (on the GPU box, from the repo root): bench line, rocprofv3 kernel stats of the same
# command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) for the conv kernels' HBM traffic.
# Everything lands under gpurun_out/final/; copy what should be judged into profiles/.
set -u
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- \
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
python tools/kstats.py $O/prof/bench_kernel_trace.csv | head -14
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- \
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > /dev/null 2> $O/pmc_$c.err
  echo "== $c (conv_mfma / conv_chain / gru)"
  python tools/pmc_summary.py $O/pmc_$c/p_counter_collection.csv conv_mfma
  python tools/pmc_summary.py $O/pmc_$c/p_counter_collection.csv conv_chain
  python tools/pmc_summary.py $O/pmc_$c/p_counter_collection.csv gru_cluster
done
