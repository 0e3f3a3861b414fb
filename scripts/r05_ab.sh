# step-level A/B on one box: tap-split weight gradient (TS 2) against the four-wave form (TS 1), two streams and serialised
set -x
O=gpurun_out/r05ab; mkdir -p $O
export L3_DEBUG_KNOBS=1
for rep in 1 2; do for ts in 1 2; do
L3_WG_TR_TS=$ts timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/ts$ts two-stream: /"
done; done
for ts in 1 2; do
L3_TWO_STREAMS=0 L3_WG_TR_TS=$ts timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 20 bf16 2>&1 | grep -a "ms/step" | head -1 | sed "s/^/ts$ts serial: /"
done
