O=gpurun_out/r05f; mkdir -p $O
export L3_DEBUG_KNOBS=1 L3_TWO_STREAMS=0
for T in base NOVALU NODMA NOMFMA NOVALU_NODMA NOVALU_NOMFMA NOEPI; do
  if [ $T = base ]; then unset L3_LIB_PATH; else export L3_LIB_PATH=scripts/probes/libl3hip_bx6_$T.so; fi
  timeout 200 python scripts/step_profile.py 64 cnn_L3_melspec2 4 f32 f2x2_bf16x6 > $O/sp_$T.txt 2>&1
  echo "$T: $(grep -a 'conv_fwd\|conv_dgrad' $O/sp_$T.txt | awk '{printf "%s %s ms  ", $1, $2}')"
done
