export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O; rm -f $O/dbg.txt
for args in "OR16 16 0 400 40" "PP24 8 0 40 60" "PP24 1 0 400 200" "PP16 1 0 400 300"; do
  timeout 900 python tools/dbg_gru.py $args 2>&1 | grep "status" | cut -c1-300 | tee -a $O/dbg.txt
done
timeout 300 python tools/gpu_debug.py timing PP16 B=1 n_steps=8 2>&1 | grep TIMING | tee -a $O/dbg.txt
