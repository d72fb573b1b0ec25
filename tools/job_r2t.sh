export TMPDIR=/tmp
O=gpurun_out/r2t; mkdir -p $O
for v in 1 2; do
for args in "PP16 B=2 n_steps=8" "PP16 B=4 n_steps=8" "PP16 B=8 n_steps=8" "OR16 B=16 n_steps=8 iters=5" "PP24 B=8 T=96000 n_steps=8 iters=5"; do
  echo "OU_GRU_V=$v $args" | tee -a $O/timings.txt
  OU_GRU_V=$v timeout 120 python tools/gpu_debug.py timing $args 2>&1 | grep -E "TIMING|Error|error" | tee -a $O/timings.txt
done
done
