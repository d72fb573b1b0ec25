export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
for bk in 0 1; do
  OU_GRU_BACKOFF=$bk timeout 300 python tools/gru_ts.py 2>&1 | grep "v2 block" | head -2 | sed "s/^/sc1_polls=$bk /" | tee -a $O/gru.txt
  OU_GRU_BACKOFF=$bk timeout 300 python tools/gpu_debug.py timing PP16 B=1 n_steps=8 2>&1 | grep TIMING | sed "s/^/sc1_polls=$bk /" | tee -a $O/gru.txt
done
timeout 900 python -m pytest tests/test_gpu_gru.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5 | tee -a $O/gru.txt
timeout 600 python tools/dbg_gru.py PP16 1 0 400 300 2>&1 | grep status | cut -c1-200 | tee -a $O/gru.txt
timeout 600 python tools/dbg_gru.py PP24 1 0 400 200 2>&1 | grep status | cut -c1-200 | tee -a $O/gru.txt
OU_TRACE=1 OU_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python tools/gpu_debug.py timing PP16 iters=2 > $O/timing_trace.txt 2> $O/trace.log
python tools/trace_summary.py $O/tr/t_kernel_trace.csv $O/trace.log > $O/layers.txt 2>&1
head -20 $O/layers.txt
