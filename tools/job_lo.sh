export TMPDIR=/tmp
O=gpurun_out/xchg; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_overhead.hip -o /tmp/lo 2> $O/build_lo.log || { cat $O/build_lo.log; exit 1; }
timeout 150 /tmp/lo 2>&1 | tee $O/launch_overhead.txt
