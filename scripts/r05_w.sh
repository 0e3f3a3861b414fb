# chunk-major filter copy for the 32-channel-chunk halo kernels
set -x
O=gpurun_out/r05w; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_bf16_stored_random_geometries or conv_layer_bf16 or forms_its_own" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed" $O/tests_a.log | tail -2
timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_flat4.txt 2>&1
cat $O/layers_flat4.txt
unset L3_DEBUG_KNOBS
timeout -k 10 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --no-secondary > $O/bench_bf16_b128_line.json 2>$O/bench_bf16.err
python -c "
import json; d=json.load(open('$O/bench_bf16_b128_line.json')); print('bf16', d['value'], d['ms_per_step'], d['value_saturated_head'], d['roofline']['frac'])"
export L3_DEBUG_KNOBS=1
timeout -k 10 1500 python -m pytest tests -q -s -m gpu -x -k "bf16 or mixed" > $O/tests_b.log 2>&1; echo "tests_b rc=$?"
grep -a "passed\|failed" $O/tests_b.log | tail -2
