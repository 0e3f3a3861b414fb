set -x
O=gpurun_out/r05d; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 1200 python -m pytest tests/test_layer_parity_gpu.py -q -s -m gpu -k "bx6 or bf16x6" > $O/tests.log 2>&1; echo "tests rc=$?"
grep -a "passed\|failed\|Error\|error" $O/tests.log | head -20
timeout 300 python scripts/step_profile.py 64 cnn_L3_melspec2 5 f32 f2x2_bf16x6 > $O/sp_bx6.txt 2>&1; tail -9 $O/sp_bx6.txt
cd /tmp && export TMPDIR=/tmp
export L3_DEBUG_KNOBS=1 L3_TWO_STREAMS=0
for A in f2x2_bf16x6; do
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$A -o t -- python $R/scripts/step_profile.py 64 cnn_L3_melspec2 4 f32 $A > $R/$O/tr_$A.log 2>&1
python $R/scripts/conv_layers_by_order.py $(find $R/$O/tr_$A -name "*kernel_trace.csv" | head -1) 28 conv_wino4_kernel conv_wino_bx6_kernel > $R/$O/layers_$A.txt
cp $(find $R/$O/tr_$A -name "*kernel_stats.csv" | head -1) $R/$O/stats_$A.csv
done
run() {  # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$O/$name -o $name -- python $R/scripts/step_profile.py 64 cnn_L3_melspec2 1 f32 f2x2_bf16x6 > $R/$O/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES
cd $R
python scripts/pmc_summarize.py $R/$O > $O/pmc_summary.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +3M -delete
cat $O/layers_f2x2_bf16x6.txt
grep -A16 "conv_wino_bx6_kernel<4, 1>" $O/pmc_summary.txt | head -40
