export TMPDIR=/tmp
O=gpurun_out/xchg; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/vmem_issue.hip -o /tmp/vmem 2> $O/build_vm.log || { cat $O/build_vm.log; exit 1; }
timeout 100 /tmp/vmem 2>&1 | tee $O/vmem_issue.txt
