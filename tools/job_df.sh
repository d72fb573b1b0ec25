export TMPDIR=/tmp
O=gpurun_out/df; mkdir -p $O
for f in 3 6 12 1000; do
for args in "PP16 B=4 n_steps=8" "PP16 B=8 n_steps=8 iters=5" "OR16 B=16 n_steps=8 iters=4"; do
  OU_DEEP_FACTOR=$f timeout 120 python tools/gpu_debug.py timing $args 2>&1 | grep -E "TIMING|Error" | sed "s/^/factor=$f /" | tee -a $O/timings.txt
done
done
