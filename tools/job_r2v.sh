export TMPDIR=/tmp
O=gpurun_out/r2v; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_load or direct_conv" 2>&1 | tail -8 | cut -c1-300
timeout 200 python tools/direct_ts.py 2>&1 | grep -v amdgpu.ids | tee $O/direct_ts.txt
for v in 1 2 1 2; do
OU_CONV_DIRECT=$v timeout 120 python tools/gpu_debug.py timing PP16 B=1 n_steps=8 2>&1 | grep -E "TIMING|Error" | sed "s/^/direct=$v /" | tee -a $O/timings.txt
done
for v in 1 2; do
OU_CONV_DIRECT=$v timeout 120 python tools/gpu_debug.py timing PP16 B=4 n_steps=8 2>&1 | grep -E "TIMING|Error" | sed "s/^/direct=$v /" | tee -a $O/timings.txt
OU_CONV_DIRECT=$v timeout 120 python tools/gpu_debug.py timing PP24 B=1 T=96000 n_steps=8 2>&1 | grep -E "TIMING|Error" | sed "s/^/direct=$v /" | tee -a $O/timings.txt
done
