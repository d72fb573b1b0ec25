export TMPDIR=/tmp
rm -f gpurun_out/failed_subprocess.txt
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 300 python -m pytest tests/test_gpu_distributed.py -m gpu -q -k bench_spawns 2>&1 | tail -1; done
grep -v "^\[Gloo\]" gpurun_out/failed_subprocess.txt | grep -B2 -A12 "Traceback\|Error" | head -80
