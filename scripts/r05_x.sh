# weight-gradient kernel at three workgroups per CU; 32-channel-chunk form (MODE 2) of the 2-D 128-channel blocks with the chunk-major filter
set -x
O=gpurun_out/r05x; mkdir -p $O
export L3_DEBUG_KNOBS=1
for occ in 2 3; do for sp in 512 768 1024; do
L3_WG_TR_OCC=$occ WG_SPLITS=$sp timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad_occ${occ}_$sp.txt 2>&1
done; done
paste -d'|' $O/wgrad_occ2_512.txt $O/wgrad_occ3_512.txt $O/wgrad_occ3_768.txt $O/wgrad_occ3_1024.txt $O/wgrad_occ2_768.txt | cut -c1-20,40-75,135-165,230-260,325-355,420-450
L3_HALO_MODE=2 timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_mode2.txt 2>&1
cat $O/layers_mode2.txt
timeout -k 10 600 python -m pytest tests -q -s -m gpu -x -k "conv_layer_bf16 or conv_bf16_stored_random_geometries or forms_its_own" > $O/tests_a.log 2>&1; echo "tests_a rc=$?"
grep -a "passed\|failed" $O/tests_a.log | tail -2
