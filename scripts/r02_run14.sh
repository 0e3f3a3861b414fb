O=gpurun_out/r02n; mkdir -p $O
for q in "" 8 16; do for p in 0 1; do echo "GPU_MAX_HW_QUEUES=$q PRIO=$p"; GPU_MAX_HW_QUEUES=$q L3_COMM_PRIO=$p timeout 300 python scripts/dp_overhead.py 64 f32 2>&1 | grep "median"; done; done | tee $O/queues.txt
