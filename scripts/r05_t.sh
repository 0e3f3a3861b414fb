# flat tiles: trimmed halo image, 4-deep ring + operands read across the barrier (MODE 4) against MODE 3 and the 2-D patches
set -x
O=gpurun_out/r05t; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_bf16_stored_random_geometries or conv_layer_bf16" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed" $O/tests_a.log | tail -2
L3_HALO_FLAT_MODE=3 timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_bf16_stored_random_geometries" > $O/tests_a3.log 2>&1; echo "tests_a3 rc=$?"
grep -a "passed\|failed" $O/tests_a3.log | tail -2
L3_HALO_FLAT=0 timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_2d.txt 2>&1
L3_HALO_FLAT_MODE=3 timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_flat3.txt 2>&1
timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_flat4.txt 2>&1
L3_HALO_FLAT=2 timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_flat4_forced.txt 2>&1
timeout 300 scripts/probes/halo_bench sweep 128 > $O/sweep.txt 2>&1
timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad.txt 2>&1
paste -d'|' $O/layers_2d.txt $O/layers_flat3.txt | cut -c1-250
cat $O/layers_flat4.txt $O/layers_flat4_forced.txt $O/sweep.txt $O/wgrad.txt
unset L3_DEBUG_KNOBS
timeout -k 10 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --no-secondary > $O/bench_bf16_b128_line.json 2>$O/bench_bf16.err
python -c "
import json; d=json.load(open('$O/bench_bf16_b128_line.json')); print(d['value'], d['ms_per_step'], d['value_saturated_head'], d['roofline']['frac'])"
