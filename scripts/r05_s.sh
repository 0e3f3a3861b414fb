# flat tiles of the bf16 halo kernel: parity tests, per-launch listing with and without them, bench
set -x
O=gpurun_out/r05s; mkdir -p $O
R=$GRAFT_REPO_ROOT
export L3_DEBUG_KNOBS=1
timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_bf16_stored_random_geometries or conv_layer_bf16 or split_tail_of_the_f4" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed" $O/tests_a.log | tail -2
cd /tmp && export TMPDIR=/tmp
for flat in 0 1; do
L3_HALO_FLAT=$flat L3_TWO_STREAMS=0 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr$flat -o bf16 -- python $R/scripts/step_profile.py 128 cnn_L3_melspec2 6 bf16 > $R/$O/tr$flat.log 2>&1
python $R/scripts/kernels_in_order.py $(find $R/$O/tr$flat -name "*kernel_trace.csv" | head -1) > $R/$O/bf16_b128_kernels_in_order_flat$flat.txt
find $R/$O/tr$flat -name "*.db" -delete; find $R/$O/tr$flat -name "*kernel_trace.csv" -delete
grep -a "ms/step" $R/$O/tr$flat.log | head -3
done
cd $R
unset L3_DEBUG_KNOBS
timeout -k 10 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --no-secondary > $O/bench_bf16_b128_line.json 2>$O/bench_bf16.err
python -c "
import json; d=json.load(open('$O/bench_bf16_b128_line.json')); print(d['value'], d['ms_per_step'], d['value_saturated_head'], d['roofline']['frac'])"
export L3_DEBUG_KNOBS=1
timeout -k 10 1500 python -m pytest tests -q -s -m gpu -x -k "bf16 or mixed" > $O/tests_b.log 2>&1; echo "tests_b rc=$?"
grep -a "passed\|failed" $O/tests_b.log | tail -2
