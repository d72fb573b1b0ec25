export TMPDIR=/tmp
O=gpurun_out/ts; mkdir -p $O
timeout 200 python tools/direct_ts.py 2>&1 | grep -v amdgpu.ids | tee $O/direct_ts.txt
