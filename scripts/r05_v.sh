# first-layer weight gradient forming its own dY (BatchNorm-backward apply deferred into it): A/B test, whole GPU suite, benches
set -x
O=gpurun_out/r05v2; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout -k 10 600 python -m pytest tests -q -s -m gpu -x -k "forms_its_own_output_gradient or first_layer_weight_gradient" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed\|Error\|assert" $O/tests_a.log | tail -8
unset L3_DEBUG_KNOBS
timeout -k 10 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --no-secondary > $O/bench_bf16_b128_line.json 2>$O/bench_bf16.err
python -c "
import json; d=json.load(open('$O/bench_bf16_b128_line.json')); print('bf16', d['value'], d['ms_per_step'], d['value_saturated_head'], d['roofline']['frac'])"
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_line.json 2>$O/bench.err
python -c "
import json; d=json.load(open('$O/bench_line.json')); print('f32', d['value'], d['ms_per_step'], d['value_saturated_head'], d['roofline']['frac'])"
timeout -k 10 2400 python -m pytest tests -q -s -m gpu -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
grep -a "passed\|failed" $O/gpu_tests.log | tail -2
