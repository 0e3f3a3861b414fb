O=gpurun_out/r05o; mkdir -p $O
export L3_DEBUG_KNOBS=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
L3_TWO_STREAMS=0 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr -o f32 -- python $R/scripts/step_profile.py 64 cnn_L3_melspec2 6 f32 > $R/$O/tr.log 2>&1
cd $R
python scripts/kernels_in_order.py $(find $O/tr -name "*kernel_trace.csv" | head -1) > $O/f32_b64_kernels_in_order.txt
tail -1 $O/f32_b64_kernels_in_order.txt
find $O/tr -name "*.db" -delete; find $O/tr -name "*kernel_trace.csv" -delete
for P in 0 1; do
L3_BNBWD_FUSE_POOLED=$P L3_TWO_STREAMS=0 timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 6 f32 2>&1 | grep -a "pairs/s\|elementwise\|conv_dgrad"
L3_BNBWD_FUSE_POOLED=$P timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 6 f32 2>&1 | grep -a "pairs/s"
L3_BNBWD_FUSE_POOLED=$P L3_TWO_STREAMS=0 timeout 200 python scripts/step_profile.py 128 cnn_L3_melspec2 6 bf16 2>&1 | grep -a "pairs/s\|elementwise\|conv_dgrad"
L3_BNBWD_FUSE_POOLED=$P timeout 200 python scripts/step_profile.py 128 cnn_L3_melspec2 6 bf16 2>&1 | grep -a "pairs/s"
done
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -x -k "golden or dgrad_epilogue or bf16 or mixed" > $O/tests.log 2>&1; grep -a "passed\|failed\|Error\|mixed-precision golden\|gradients: worst" $O/tests.log | tail -8
