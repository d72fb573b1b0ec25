export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
OU_TRACE=1 OU_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python tools/gpu_debug.py timing PP16 iters=2 > $O/timing_trace.txt 2> $O/trace.log
python tools/trace_summary.py $O/tr/t_kernel_trace.csv $O/trace.log > $O/layers.txt 2>&1
tail -5 $O/layers.txt
for args in "PP16 B=1 n_steps=8" "PP16 B=4 n_steps=64 iters=3" "PP16 B=8 n_steps=8" "OR16 B=16 n_steps=32 iters=3" "PP24 B=8 T=96000 n_steps=8 iters=3" "PP24 B=1 T=96000 n_steps=8"; do
  timeout 600 python tools/gpu_debug.py timing $args 2>&1 | grep TIMING | tee -a $O/timings.txt
done
