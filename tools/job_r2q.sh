export TMPDIR=/tmp
O=gpurun_out/r2q; mkdir -p $O
for d in 0 1 2 4 7; do
OU_DBG=$d OU_TRACE=1 OU_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr$d -o t -- python tools/gpu_debug.py timing PP16 iters=2 > $O/timing_trace.txt 2> $O/trace$d.log
echo "== dbg $d"; python tools/trace_summary.py $O/tr$d/t_kernel_trace.csv $O/trace$d.log | grep "score.*\.up "
done
