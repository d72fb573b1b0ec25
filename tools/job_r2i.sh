export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
for u in 16 8; do
  OU_GRU_UPW=$u timeout 300 python tools/gru_ts.py 2>&1 | grep "v2 block" | head -2 | sed "s/^/upw=$u /" | tee -a $O/gru.txt
  OU_GRU_UPW=$u timeout 300 python tools/gpu_debug.py timing PP16 B=1 n_steps=8 2>&1 | grep TIMING | sed "s/^/upw=$u /" | tee -a $O/gru.txt
done
timeout 600 python -m pytest tests/test_gpu_gru.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O/gru.txt
OU_GRU_UPW=8 timeout 600 python -m pytest tests/test_gpu_gru.py -m gpu -x -q -k "PP16-1 or PP16-2" 2>&1 | tail -3 | tee -a $O/gru.txt
