# Same-box A/B of two builds of libl3hip.so: bash scripts/ab_step.sh <out dir> <lib a> <lib b> [rounds]
O=$1; A=$2; B=$3; R=${4:-3}; mkdir -p $O
export L3_DEBUG_KNOBS=1
for r in $(seq 1 $R); do
  for lib in $A $B; do
    n=$(basename $lib .so)
    L3_LIB_PATH=$lib timeout 600 python scripts/step_profile.py 64 cnn_L3_melspec2 20 > $O/${n}_$r.txt 2>&1
    echo "$n round $r: $(grep 'ms/step,' $O/${n}_$r.txt)  | $(grep -E 'conv_fwd|conv_dgrad' $O/${n}_$r.txt | awk '{printf "%s %s  ", $1, $2}')"
  done
done
