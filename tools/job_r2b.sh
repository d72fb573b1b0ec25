export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout 300 python tools/conv_ts.py > $O/conv_ts.txt 2>&1
timeout 300 python tools/conv_ablate.py > $O/conv_ablate.txt 2>&1
timeout 300 python tools/gru_ts.py > $O/gru_ts.txt 2>&1
cat $O/conv_ts.txt $O/conv_ablate.txt $O/gru_ts.txt
