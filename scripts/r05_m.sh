O=gpurun_out/r05m; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout 600 python -m pytest tests/test_layer_parity_gpu.py -q -s -m gpu -k "split_bf16_experiment" > $O/wgbx6_tests.log 2>&1; echo rc=$?
grep -a "passed\|failed" $O/wgbx6_tests.log | tail -2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
L3_TWO_STREAMS=0 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr -o bf16 -- python $R/scripts/step_profile.py 128 cnn_L3_melspec2 6 bf16 > $R/$O/tr.log 2>&1
cd $R
python scripts/kernels_in_order.py $(find $O/tr -name "*kernel_trace.csv" | head -1) > $O/bf16_b128_kernels_in_order.txt
tail -1 $O/bf16_b128_kernels_in_order.txt
grep -a "ms/step" $O/tr.log | head -20
find $O/tr -name "*.db" -delete; find $O/tr -name "*kernel_trace.csv" -delete
bash scripts/pmc_bf16.sh $O/pmc 128 > $O/pmc.log 2>&1
grep -a -A24 "== conv_bf16_halo_kernel<32, 2, 1, true, 1>" $O/pmc/summary.txt | cut -c1-100
