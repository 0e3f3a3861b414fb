#!/bin/bash
# PMC passes for the bf16 (mixed-precision) kernels; counters in their own runs (--pmc with --kernel-trace only).
# usage: scripts/pmc_bf16.sh <outdir> [batch]
set -u
OUT=${1:-gpurun_out/pmc_bf16}
B=${2:-128}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export L3_DEBUG_KNOBS=1 L3_TWO_STREAMS=0
R=$GRAFT_REPO_ROOT
run() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$name -o $name -- python $R/scripts/step_profile.py $B cnn_L3_melspec2 1 bf16 > $R/$OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES
run sq3 SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
cd $R && python $R/scripts/pmc_summarize.py $R/$OUT > $R/$OUT/summary.txt 2>&1
find $R/$OUT -name "*kernel_trace.csv" -delete
find $R/$OUT -name "*counter_collection.csv" -size +3M -delete
find $R/$OUT -name "*.db" -delete
