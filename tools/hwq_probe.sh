#!/bin/bash
# Two open questions of round 4: why `GPU_MAX_HW_QUEUES=2` hangs bench.py, and why three lanes are slower than two
# on ragged sets.  Every step under its own timeout; a hang is reported, not waited for.
export TMPDIR=/tmp
O=${1:-gpurun_out/hwq}; mkdir -p $O
BQ="--steps 3 --warmup 1 --no-cpu-baseline --sustained-s 0 --in-flight= --batch-sweep= --profile-steps 1"
probe() {  # name, env..., --, command...
  local name=$1; shift
  local t0=$SECONDS
  if env "$@" > $O/$name.out 2> $O/$name.err; then rc=ok; else rc="rc=$?"; fi
  printf "%-44s %-8s %4d s  %s\n" "$name" "$rc" "$((SECONDS - t0))" "$(grep -o '"ms_per_step": [0-9.]*' $O/$name.out | head -1)"
}
{
echo "== GPU_MAX_HW_QUEUES = 2 (rc=124: killed by the 90 s timeout, rc=139: segmentation fault)"
probe hwq2_bench_default        GPU_MAX_HW_QUEUES=2 timeout 90 python bench.py $BQ
probe hwq2_bench_no_overlap     GPU_MAX_HW_QUEUES=2 timeout 90 python bench.py --option no_overlap=1 $BQ
probe hwq2_bench_no_graph       GPU_MAX_HW_QUEUES=2 OU_BENCH_NO_GRAPH=1 timeout 90 python bench.py $BQ
probe hwq1_bench_no_graph       GPU_MAX_HW_QUEUES=1 OU_BENCH_NO_GRAPH=1 timeout 90 python bench.py $BQ
probe hwq4_bench_default        GPU_MAX_HW_QUEUES=4 timeout 90 python bench.py $BQ
echo "== where the default run of GPU_MAX_HW_QUEUES=2 stops (faulthandler dump of the host thread after 40 s)"
GPU_MAX_HW_QUEUES=2 timeout 90 python -X faulthandler -c "
import faulthandler, sys; faulthandler.dump_traceback_later(40, exit=True)
sys.argv=['bench.py'] + '$BQ'.split()
import runpy; runpy.run_path('bench.py', run_name='__main__')" > $O/hwq2_where.out 2> $O/hwq2_where.err
grep -E 'File \"[^\"]*(bench|universe|graphs)\.py' $O/hwq2_where.err | head -8
} 2>&1 | tee $O/hwq_probe.txt
{
echo "== three lanes on ragged sets (tools/lanes_rate.py, ragged 3.5-4.0 s, 32 utterances)"
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 600 python tools/lanes_rate.py PP16 32 1,2,3,4 2>&1 | grep -E "ragged 3.5|GPU_MAX"
done
unset GPU_MAX_HW_QUEUES
for K in 2 3 4; do timeout 300 python tools/lanes_timeline.py $K 32 2>&1 | grep -v amdgpu.ids; done
} 2>&1 | tee $O/lanes3_probe.txt
