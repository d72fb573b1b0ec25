export TMPDIR=/tmp
O=gpurun_out/r2u; mkdir -p $O
rm -f gpurun_out/parity_observed.json gpurun_out/failed_subprocess.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt | cut -c1-300
for args in "PP16 B=1 n_steps=8" "PP16 B=4 n_steps=8" "PP24 B=1 T=96000 n_steps=8" "OR16 B=1 n_steps=8"; do
  timeout 200 python tools/gpu_debug.py timing $args 2>&1 | grep TIMING | tee -a $O/timings.txt
done
