set -x
O=gpurun_out/r02e; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/tr_probe scripts/probes/tr_b16_probe.hip 2>/dev/null && /tmp/tr_probe > $O/tr_probe.txt 2>&1
for abl in 0 1 2 3 4 8 12 15; do
  L3_TWO_STREAMS=0 L3_HALO_ABL=$abl timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 3 bf16 2>/dev/null | grep "conv_fwd\|conv_dgrad\|ms/step" > $O/abl_$abl.txt
done
L3_TWO_STREAMS=0 L3_BF16_HALO=0 timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 3 bf16 2>/dev/null | grep "conv_fwd\|conv_dgrad\|ms/step" > $O/abl_nohalo.txt
bash scripts/pmc_bf16.sh $O/pmc 128 > $O/pmc.log 2>&1
head -50 $O/pmc/summary.txt
for f in $O/abl_*.txt; do echo $f; cat $f; done
