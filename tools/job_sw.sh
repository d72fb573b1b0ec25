export TMPDIR=/tmp
O=gpurun_out/sweep; mkdir -p $O
timeout 300 python tools/direct_sweep.py PP16 1 2>&1 | grep -v amdgpu.ids | tee $O/direct_sweep_B1.txt
