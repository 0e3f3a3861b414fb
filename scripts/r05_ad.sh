# first-layer weight gradient with the ReLU mode as a template parameter
set -x
O=gpurun_out/r05ad; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "forms_its_own or first_layer_weight_gradient or training_step_matches_golden or conv_bf16_stored_random_geometries" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed" $O/tests_a.log | tail -2
unset L3_DEBUG_KNOBS
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_line.json 2>$O/bench.err
python -c "
import json; d=json.load(open('$O/bench_line.json')); print('f32', d['value'], d['ms_per_step'], d['value_saturated_head'], d['roofline']['frac'])"
timeout -k 10 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --no-secondary > $O/bench_bf16_b128_line.json 2>$O/bench_bf16.err
python -c "
import json; d=json.load(open('$O/bench_bf16_b128_line.json')); print('bf16', d['value'], d['ms_per_step'], d['value_saturated_head'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_f32 -o f32 -- python $R/bench.py --serial --steps 10 --no-cpu-baseline --no-secondary --no-saturated --roofline-steps 0 > $R/$O/prof_f32.log 2>&1
grep -a "first_wgrad" $(find $R/$O/prof_f32 -name "*kernel_stats.csv" | head -1) | cut -c1-120
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bf16 -o bf16 -- python $R/bench.py --serial --steps 10 --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --no-secondary --no-saturated --roofline-steps 0 > $R/$O/prof_bf16.log 2>&1
grep -a "first_wgrad" $(find $R/$O/prof_bf16 -name "*kernel_stats.csv" | head -1) | cut -c1-120
find $R/$O -name "*.db" -delete; find $R/$O -name "*kernel_trace.csv" -delete
