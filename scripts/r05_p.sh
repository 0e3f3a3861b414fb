O=gpurun_out/r05p; mkdir -p $O
export L3_DEBUG_KNOBS=1
for P in 0 1 0 1; do
echo "L3_WEIGHT_STREAM=$P"
L3_WEIGHT_STREAM=$P timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 10 f32 2>&1 | grep -a "pairs/s"
done
L3_TWO_STREAMS=0 timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 6 f32 2>&1 | grep -a "pairs/s\|elementwise\|conv_"
L3_TWO_STREAMS=0 timeout 200 python scripts/step_profile.py 128 cnn_L3_melspec2 6 bf16 2>&1 | grep -a "pairs/s\|elementwise"
timeout 200 python scripts/step_profile.py 128 cnn_L3_melspec2 6 bf16 2>&1 | grep -a "pairs/s"
timeout 2400 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -x > $O/tests.log 2>&1; grep -a "passed\|failed\|Error" $O/tests.log | tail -5
