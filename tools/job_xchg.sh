export TMPDIR=/tmp
O=gpurun_out/xchg; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/xchg_latency.hip -o /tmp/xchg 2> $O/build.log || { cat $O/build.log; exit 1; }
timeout 150 /tmp/xchg 2>&1 | tee $O/xchg.txt
