export TMPDIR=/tmp
O=gpurun_out/nc; mkdir -p $O
for nc in 0 128 256; do
OU_FUSE_NC=$nc timeout 120 python tools/gpu_debug.py timing PP16 B=1 n_steps=8 2>&1 | grep -E "TIMING|Error" | sed "s/^/nc=$nc /" | tee -a $O/timings.txt
done
for f in 0 2 3; do
OU_FUSE=$f timeout 120 python tools/gpu_debug.py timing PP16 B=1 n_steps=8 2>&1 | grep -E "TIMING|Error" | sed "s/^/fuse=$f /" | tee -a $O/timings.txt
done
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/nc/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('ms', d['ms_per_step'], d['value'])
print(r['kernel'][:30], r['achieved'], r['frac'], r['avg_launch_us'], r['launches'])
for k,v in r['other_conv_kernels'].items(): print('   ',k[:40], round(v['achieved'],1), round(v['avg_launch_us'],1), v['launches'])
PY
