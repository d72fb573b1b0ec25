export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
timeout 600 python tools/gpu_debug.py timing PP24 B=8 T=96000 n_steps=8 iters=3 > $O/pp24_b8.txt 2>&1; tail -5 $O/pp24_b8.txt
timeout 1500 python -m pytest tests/test_gpu_gru.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for d in 0 1; do
  for args in "PP16 B=1 n_steps=8" "PP16 B=4 n_steps=8" "PP24 B=1 T=96000 n_steps=8"; do
    OU_CONV_DIRECT=$d timeout 600 python tools/gpu_debug.py timing $args 2>&1 | grep TIMING | sed "s/^/direct=$d /" | tee -a $O/timings.txt
  done
done
OU_TRACE=1 OU_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python tools/gpu_debug.py timing PP16 iters=2 > $O/timing_trace.txt 2> $O/trace.log
python tools/trace_summary.py $O/tr/t_kernel_trace.csv $O/trace.log > $O/layers.txt 2>&1
head -30 $O/layers.txt
