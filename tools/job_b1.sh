export TMPDIR=/tmp
O=gpurun_out/b1; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b1/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['free_running'], d['host_enqueue'])
r=d['roofline']; print(r['kernel'], r['achieved'], r['frac'], r['avg_launch_us'], r['launches'])
for k,v in r['other_conv_kernels'].items(): print(k[:30], v['achieved'], v['avg_launch_us'], v['launches'])
print(r['score_forward']); print(d['cpu_baseline'])
PY
