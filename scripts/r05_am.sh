# what the operand DATA costs: the same launches on all-zero tensors (power / clock)
set -x
O=gpurun_out/r05am; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_rand.txt 2>&1
HB_DATA=zero timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_zero.txt 2>&1
timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad_rand.txt 2>&1
HB_DATA=zero timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad_zero.txt 2>&1
paste -d'|' $O/layers_rand.txt $O/layers_zero.txt | cut -c1-34,35-50,78-92,130-147,175-190
for f in wgrad_rand wgrad_zero; do echo $f; grep -a "wgrad (" $O/$f.txt | sed 's/.*splits) *//' | awk '{printf "%s ", $1}'; echo; done
