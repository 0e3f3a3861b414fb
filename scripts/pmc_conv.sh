#!/bin/bash
# PMC passes for the conv kernels (counters in their own runs: --pmc with --kernel-trace only).
# usage: scripts/pmc_conv.sh <outdir> [wino4_layer_durations.txt of scripts/wino_layers_by_order.py: adds traffic_by_layer.txt]
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export L3_DEBUG_KNOBS=1 L3_TWO_STREAMS=0   # per-kernel counters: towers serialised
R=$GRAFT_REPO_ROOT
run() {  # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$name -o $name -- python $R/scripts/step_profile.py 64 cnn_L3_melspec2 1 > $R/$OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
cd $R && python - <<'PY'
import csv, glob, os, sys, collections
out = os.environ.get('OUT', sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc')
PY
python $R/scripts/pmc_summarize.py $R/$OUT > $R/$OUT/summary.txt 2>&1
cat $R/$OUT/summary.txt
# per-layer HBM traffic of the 28 F(4x4,3x3) launches (needs the per-layer durations of a kernel trace of the same workload)
if [ -n "${2:-}" ] && [ -f "$R/$2" ]; then python $R/scripts/traffic_by_layer.py $R/$OUT $R/$2 > $R/$OUT/traffic_by_layer.txt 2>&1; fi
find $R/$OUT -name "*kernel_trace.csv" -delete
find $R/$OUT -name "*counter_collection.csv" -size +3M -delete
