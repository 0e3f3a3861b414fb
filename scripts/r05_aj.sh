# weight gradient with the next patch's pieces spread behind the k-steps (L3_WG_TR_SP=1)
set -x
O=gpurun_out/r05aj; mkdir -p $O
export L3_DEBUG_KNOBS=1
L3_WG_TR_SP=1 timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_bf16_stored_random_geometries and (halo_auto or wgrad_8x8 or wgrad_4x16)" > $O/tests_sp.log 2>&1; echo "tests_sp rc=$?"
grep -a "passed\|failed" $O/tests_sp.log | tail -1
L3_WG_TR_SP=1 timeout -k 10 900 python -m pytest tests -q -s -m gpu -x -k "conv_layer_bf16" > $O/tests_sp2.log 2>&1; echo "tests_sp2 rc=$?"
grep -a "passed\|failed" $O/tests_sp2.log | tail -1
timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad_base.txt 2>&1
L3_WG_TR_SP=1 timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad_sp.txt 2>&1
for f in wgrad_base wgrad_sp; do echo $f; grep -a "wgrad (" $O/$f.txt | sed 's/.*splits) *//' | awk '{printf "%s ", $1}'; echo; done
for rep in 1 2; do for sp in 0 1; do
L3_WG_TR_SP=$sp timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/sp=$sp two-stream: /"
done; done
