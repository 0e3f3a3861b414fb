# co-residency experiment: halo blocks at one per CU (81 KiB of LDS each) / three 64-channel blocks per CU (54 KiB), step with two streams and serialised
set -x
export L3_DEBUG_KNOBS=1
for rep in 1 2; do for m in 0 81 54; do
L3_HALO_LDS_MIN=$m timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/lds_min $m two-stream: /"
done; done
for m in 0 81; do
L3_TWO_STREAMS=0 L3_HALO_LDS_MIN=$m timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/lds_min $m serial: /"
done
