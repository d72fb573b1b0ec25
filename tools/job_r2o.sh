export TMPDIR=/tmp
O=gpurun_out/r2o; mkdir -p $O
timeout 120 python tools/gpu_debug.py timing PP16 B=1 n_steps=8 2>&1 | grep -E "TIMING|Error" | tee -a $O/timings.txt
OU_FUSE_UPFIR=0 timeout 120 python tools/gpu_debug.py timing PP16 B=1 n_steps=8 2>&1 | grep -E "TIMING|Error" | tee -a $O/timings.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_up or folded_fir or direct_conv" 2>&1 | tail -15 | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt | cut -c1-300
for args in "PP16 B=4 n_steps=8" "PP24 B=1 T=96000 n_steps=8" "OR16 B=1 n_steps=8"; do
  timeout 200 python tools/gpu_debug.py timing $args 2>&1 | grep TIMING | tee -a $O/timings.txt
done
OU_TRACE=1 OU_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python tools/gpu_debug.py timing PP16 iters=2 > $O/timing_trace.txt 2> $O/trace.log
python tools/trace_summary.py $O/tr/t_kernel_trace.csv $O/trace.log > $O/layers.txt 2>&1
head -26 $O/layers.txt; grep "\.up" $O/layers.txt
