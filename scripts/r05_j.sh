O=gpurun_out/r05j; mkdir -p $O
for P in 1 0; do
L3_DEBUG_KNOBS=1 L3_WINO_PERSIST=$P L3_RCCL_LIB=tests/fake_rccl/libfake_rccl.so FAKE_RCCL_DELAY_US=300 GPU_MAX_HW_QUEUES=8 python tests/dp_fake_worker.py overlap cnn_L3_melspec2 64 20 2 | grep RESULT > $O/overlap_persist$P.txt
echo "persist=$P $(cat $O/overlap_persist$P.txt | cut -c1-120)"
done
timeout -k 10 2400 python -m pytest tests -q -x -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/gpu_tests.log
