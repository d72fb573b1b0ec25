export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O; rm -f $O/dbg.txt
timeout 300 python tools/gpu_debug.py full PP16 4 64000 2 2>&1 | grep -v amdgpu.ids | tee -a $O/dbg.txt
timeout 200 python tools/gpu_debug.py full PP16 2 12000 2 2>&1 | grep -v amdgpu.ids | tee -a $O/dbg.txt
