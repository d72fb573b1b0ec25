export TMPDIR=/tmp
O=gpurun_out/r2w; mkdir -p $O
rm -f gpurun_out/parity_observed.json
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt | cut -c1-300
for args in "PP16 B=1 n_steps=8" "PP16 B=4 n_steps=8" "PP24 B=1 T=96000 n_steps=8" "OR16 B=1 n_steps=8"; do
  timeout 200 python tools/gpu_debug.py timing $args 2>&1 | grep TIMING | tee -a $O/timings.txt
done
OU_TRACE=1 OU_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python tools/gpu_debug.py timing PP16 iters=2 > $O/timing_trace.txt 2> $O/trace.log
python tools/trace_summary.py $O/tr/t_kernel_trace.csv $O/trace.log > $O/layers.txt 2>&1
head -32 $O/layers.txt; grep "score\." $O/layers.txt
