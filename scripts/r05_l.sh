O=gpurun_out/r05l; mkdir -p $O
export L3_DEBUG_KNOBS=1
L3_WG_BX6=1 timeout 900 python -m pytest tests/test_layer_parity_gpu.py -q -s -m gpu -k "test_conv_layer_fp32 and product" > $O/tests.log 2>&1; echo rc=$?
grep -a " product \|passed\|failed" $O/tests.log | cut -c1-140 | head -24
export L3_TWO_STREAMS=0
L3_WG_BX6=0 timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 4 f32 f4x4 > $O/sp_wg_fp32.txt 2>&1; grep -a "conv_wgrad\|pairs/s" $O/sp_wg_fp32.txt
L3_WG_BX6=1 timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 4 f32 f4x4 > $O/sp_wg_bx6.txt 2>&1; grep -a "conv_wgrad\|pairs/s" $O/sp_wg_bx6.txt
unset L3_TWO_STREAMS
L3_WG_BX6=0 timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 6 f32 f4x4 2>&1 | grep -a "pairs/s"
L3_WG_BX6=1 timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 6 f32 f4x4 2>&1 | grep -a "pairs/s"
