# three patch stages in the bf16 weight gradient
set -x
O=gpurun_out/r05ak; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_bf16_stored_random_geometries or conv_layer_bf16" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed" $O/tests_a.log | tail -1
timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad.txt 2>&1
L3_WG_TR_TS=2 timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad_ts2.txt 2>&1
for f in wgrad wgrad_ts2; do echo $f; grep -a "wgrad (" $O/$f.txt | sed 's/.*splits) *//' | awk '{printf "%s ", $1}'; echo; done
for rep in 1 2; do
timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/three stages two-stream: /"
done
timeout -k 10 1500 python -m pytest tests -q -s -m gpu -x -k "bf16 or mixed" > $O/tests_b.log 2>&1; echo "tests_b rc=$?"
grep -a "passed\|failed" $O/tests_b.log | tail -1
