O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -k "slow_collectives or fake_collective or keras_order or rccl_world1" > $O/tests.log 2>&1; echo rc=$?
grep -a "plain step\|passed\|failed\|Error\|assert" $O/tests.log | head
L3_DEBUG_KNOBS=1 L3_DP_ARENA_ORDER=1 L3_RCCL_LIB=tests/fake_rccl/libfake_rccl.so FAKE_RCCL_DELAY_US=300 GPU_MAX_HW_QUEUES=8 python tests/dp_fake_worker.py overlap cnn_L3_melspec2 64 20 2 | grep RESULT > $O/overlap_arena_order.txt
L3_DEBUG_KNOBS=1 L3_RCCL_LIB=tests/fake_rccl/libfake_rccl.so FAKE_RCCL_DELAY_US=300 GPU_MAX_HW_QUEUES=8 python tests/dp_fake_worker.py overlap cnn_L3_melspec2 64 20 2 | grep RESULT > $O/overlap_alternating.txt
cat $O/overlap_arena_order.txt $O/overlap_alternating.txt
timeout 300 python bench.py --force-comm --no-cpu-baseline --no-secondary --no-saturated --steps 20 --warmup 5 2>/dev/null > $O/line_forcecomm.json
python -c "
import json; d=json.load(open('$O/line_forcecomm.json')); print(d['value'], d['comm'])"
