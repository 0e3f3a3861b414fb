set -x
O=gpurun_out/r02f; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -k "bf16_stored_random or bf16_training or full_size or conv_random" > $O/halo.log 2>&1; echo "halo rc=$?"
timeout 1500 python -m pytest tests/test_layer_parity_gpu.py -q -s -m gpu -k "bf16" > $O/layer.log 2>&1; echo "layer rc=$?"
timeout 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline > $O/bench_bf16_b128.json 2> $O/bench_bf16.err; echo "bench16 rc=$?"
L3_TWO_STREAMS=0 L3_PROFILE_VERBOSE=1 timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 3 bf16 > $O/step_profile_bf16.log 2>&1
tail -3 $O/halo.log; tail -3 $O/layer.log; grep "ms/step" $O/step_profile_bf16.log
