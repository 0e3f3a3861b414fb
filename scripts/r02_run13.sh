O=gpurun_out/r02m; mkdir -p $O
timeout 300 python scripts/dp_overhead.py 64 f32 2>&1 | grep "median\|resident" > $O/dp_skip.txt
L3_COMM_W1_CALL=1 timeout 300 python scripts/dp_overhead.py 64 f32 2>&1 | grep "median\|resident" > $O/dp_call.txt
cat $O/dp_skip.txt $O/dp_call.txt
