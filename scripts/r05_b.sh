set -x
O=gpurun_out/r05b; mkdir -p $O
timeout -k 10 1200 python -m pytest tests/test_layer_parity_gpu.py -q -s -m gpu -k "bx6 or bf16x6" > $O/tests.log 2>&1; echo "tests rc=$?"
grep -a "bx6 edge\|dx err\|f2x2_bf16x6 \|passed\|failed\|Error\|error" $O/tests.log | head -60
timeout 300 python scripts/step_profile.py 64 cnn_L3_melspec2 5 f32 f4x4 > $O/sp_f4x4.txt 2>&1; tail -9 $O/sp_f4x4.txt
timeout 300 python scripts/step_profile.py 64 cnn_L3_melspec2 5 f32 f2x2_bf16x6 > $O/sp_bx6.txt 2>&1; tail -9 $O/sp_bx6.txt
timeout 300 python scripts/step_profile.py 64 cnn_L3_melspec2 5 f32 f2x2 > $O/sp_f2x2.txt 2>&1; tail -9 $O/sp_f2x2.txt
