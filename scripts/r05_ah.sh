# flat tiles MODE 5 (64-B swizzled halo rows, both halo buffers) against MODE 4
set -x
O=gpurun_out/r05ah; mkdir -p $O
export L3_DEBUG_KNOBS=1
for m in 5 4; do
L3_HALO_FLAT_MODE=$m timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_bf16_stored_random_geometries and (halo_flat or halo_auto)" > $O/tests_m$m.log 2>&1; echo "tests mode $m rc=$?"
grep -a "passed\|failed" $O/tests_m$m.log | tail -1
done
timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_layer_bf16 or conv_bf16_stored_random_geometries" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed" $O/tests_a.log | tail -1
L3_HALO_FLAT_MODE=4 timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_m4.txt 2>&1
timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_auto.txt 2>&1
paste -d'|' $O/layers_m4.txt $O/layers_auto.txt | cut -c1-34,35-50,78-92,130-147,175-190
for rep in 1 2; do for m in 4 x; do
if [ $m = x ]; then unset L3_HALO_FLAT_MODE; else export L3_HALO_FLAT_MODE=4; fi
timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/flat mode $m two-stream: /"
done; done
unset L3_HALO_FLAT_MODE
timeout -k 10 1500 python -m pytest tests -q -s -m gpu -x -k "bf16 or mixed" > $O/tests_b.log 2>&1; echo "tests_b rc=$?"
grep -a "passed\|failed" $O/tests_b.log | tail -1
