# tap-split (8-wave) form of the bf16 weight-gradient kernel
set -x
O=gpurun_out/r05aa; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_bf16_stored_random_geometries or conv_layer_bf16" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed" $O/tests_a.log | tail -2
L3_WG_TR_TS=1 timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad_ts1.txt 2>&1
timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad_ts2.txt 2>&1
WG_SPLITS=1024 timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad_ts2_1024.txt 2>&1
for f in wgrad_ts1 wgrad_ts2 wgrad_ts2_1024; do echo $f; awk '{print $1, $6}' $O/$f.txt | tr '\n' ';'; echo; done
unset L3_DEBUG_KNOBS
timeout -k 10 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --no-secondary > $O/bench_bf16_b128_line.json 2>$O/bench_bf16.err
python -c "
import json; d=json.load(open('$O/bench_bf16_b128_line.json')); print('bf16', d['value'], d['ms_per_step'], d['value_saturated_head'], d['roofline']['frac'], d['kernels']['conv_wgrad']['frac'], d['kernels']['conv_wgrad']['ms_per_step'])"
export L3_DEBUG_KNOBS=1
timeout -k 10 1500 python -m pytest tests -q -s -m gpu -x -k "bf16 or mixed" > $O/tests_b.log 2>&1; echo "tests_b rc=$?"
grep -a "passed\|failed" $O/tests_b.log | tail -2
