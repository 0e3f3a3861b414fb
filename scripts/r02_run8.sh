set -x
O=gpurun_out/r02h; mkdir -p $O
timeout 2400 python -m pytest tests -q -s -m gpu -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
timeout 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline > $O/bench_bf16_b128.json 2> $O/bench_bf16.err; echo "bench16 rc=$?"
timeout 600 python bench.py --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench32 rc=$?"
L3_CONV_FIRST=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_f32_nofirst.json 2>/dev/null
L3_TWO_STREAMS=0 timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 3 bf16 2>/dev/null | grep "ms/step" > $O/step_profile_bf16.txt
L3_TWO_STREAMS=0 timeout 300 python scripts/step_profile.py 64 cnn_L3_melspec2 3 f32 2>/dev/null | grep "ms/step" > $O/step_profile_f32.txt
tail -4 $O/gpu_tests.log; cat $O/step_profile_bf16.txt $O/step_profile_f32.txt
