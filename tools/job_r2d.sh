export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gru.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_gru.txt 2>&1
tail -15 $O/pytest_gru.txt
for v in 1 2; do
  OU_GRU_V=$v timeout 300 python tools/gru_ts.py > $O/gru_ts_v$v.txt 2>&1; tail -4 $O/gru_ts_v$v.txt
  for args in "PP16 B=1 n_steps=8" "PP16 B=8 n_steps=8" "OR16 B=16 n_steps=32 iters=3" "PP24 B=8 T=96000 n_steps=8 iters=3"; do
    OU_GRU_V=$v timeout 600 python tools/gpu_debug.py timing $args 2>&1 | grep TIMING | sed "s/^/v$v /" | tee -a $O/timings.txt
  done
done
