# static priority for the second half of a halo block's waves (L3_HALO_PRIO=1)
set -x
O=gpurun_out/r05al; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_p0.txt 2>&1
L3_HALO_PRIO=1 timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_p1.txt 2>&1
paste -d'|' $O/layers_p0.txt $O/layers_p1.txt | cut -c1-34,35-50,78-92,130-147,175-190
for rep in 1 2; do for p in 0 1; do
L3_HALO_PRIO=$p timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/prio=$p two-stream: /"
done; done
