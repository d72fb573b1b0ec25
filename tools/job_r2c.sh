export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
rm -f gpurun_out/parity_observed.json
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.txt 2>&1
tail -30 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
