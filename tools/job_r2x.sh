export TMPDIR=/tmp
O=gpurun_out/r2x; mkdir -p $O
rm -f gpurun_out/parity_observed.json gpurun_out/failed_subprocess.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
