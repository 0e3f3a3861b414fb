O=gpurun_out/r05q; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -x -k "slow_collectives or golden or dgrad_epilogue or tower_step" > $O/tests.log 2>&1; grep -a "passed\|failed\|Error\|plain step" $O/tests.log | cut -c1-330 | tail -6
L3_BNBWD_FUSE_POOLED=0 timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -x -k "slow_collectives" > $O/tests2.log 2>&1; grep -a "passed\|failed\|Error\|plain step" $O/tests2.log | cut -c1-330 | tail -4
timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 10 f32 2>&1 | grep -a "pairs/s"
