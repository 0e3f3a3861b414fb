# step-level A/B on one box: flat tiles MODE 4 (120 registers) against MODE 3 (112: room for the other tower's BatchNorm waves)
set -x
export L3_DEBUG_KNOBS=1
for rep in 1 2; do for m in 4 3; do
L3_HALO_FLAT_MODE=$m timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/flat mode $m two-stream: /"
done; done
for m in 4 3; do
L3_TWO_STREAMS=0 L3_HALO_FLAT_MODE=$m timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/flat mode $m serial: /"
done
