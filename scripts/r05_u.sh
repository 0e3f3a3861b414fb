set -x
O=gpurun_out/r05u; mkdir -p $O
export L3_DEBUG_KNOBS=1
timeout 300 scripts/probes/halo_bench sweep 32 20 > $O/sweep.txt 2>&1
cat $O/sweep.txt
timeout 300 scripts/probes/halo_bench wgrad 128 > $O/wgrad.txt 2>&1
cat $O/wgrad.txt
