export TMPDIR=/tmp
O=gpurun_out/gts; mkdir -p $O
timeout 200 python tools/gru_ts.py 2>&1 | grep -v amdgpu.ids | tee $O/gru_ts.txt
