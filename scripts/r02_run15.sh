O=gpurun_out/r02o; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep "passed\|failed\|FAILED" $O/gpu_tests.log | tail -8
timeout 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline > $O/bench_bf16_b128.json 2> $O/bench_bf16.err; echo "bench16 rc=$?"
timeout 600 python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench32 rc=$?"
L3_TWO_STREAMS=0 timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 3 bf16 2>/dev/null | grep "ms/step" > $O/step_profile_bf16.txt; cat $O/step_profile_bf16.txt
L3_TWO_STREAMS=0 timeout 300 python scripts/step_profile.py 64 cnn_L3_melspec2 3 f32 2>/dev/null | grep "ms/step" > $O/step_profile_f32.txt; cat $O/step_profile_f32.txt
