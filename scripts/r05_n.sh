O=gpurun_out/r05n; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -x -k "golden or first or bn_ or batchnorm or frontend or input_bn or mixed" > $O/tests.log 2>&1; echo rc=$?
grep -a "passed\|failed\|mixed-precision golden\|activations, mean\|gradients: worst\|err/rms" $O/tests.log | cut -c1-200 | tail -30
export L3_TWO_STREAMS=0
timeout 200 python scripts/step_profile.py 128 cnn_L3_melspec2 6 bf16 > $O/sp_bf16.txt 2>&1; grep -a "ms/step\|pairs/s" $O/sp_bf16.txt
timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 6 f32 > $O/sp_f32.txt 2>&1; grep -a "ms/step\|pairs/s" $O/sp_f32.txt
unset L3_TWO_STREAMS
timeout 200 python scripts/step_profile.py 128 cnn_L3_melspec2 6 bf16 2>&1 | grep -a "pairs/s"
timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 6 f32 2>&1 | grep -a "pairs/s"
