O=gpurun_out/r05g; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export L3_DEBUG_KNOBS=1 L3_TWO_STREAMS=0
for T in ${VARIANTS:-base NOVALU_NOMFMA NOVALU_NODMA NOEPI NOLOOP}; do
  if [ $T = base ]; then unset L3_LIB_PATH; else export L3_LIB_PATH=$R/scripts/probes/libl3hip_bx6_$T.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_$T -o t -- python $R/scripts/step_profile.py 64 cnn_L3_melspec2 3 f32 f2x2_bf16x6 > $R/$O/tr_$T.log 2>&1
  python $R/scripts/conv_layers_by_order.py $(find $R/$O/tr_$T -name "*kernel_trace.csv" | head -1) 28 conv_wino_bx6_kernel > $R/$O/layers_$T.txt
done
cd $R
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import re,glob,os
O='gpurun_out/r05g'
cols={}
for f in sorted(glob.glob(O+'/layers_*.txt')):
    t=os.path.basename(f)[7:-4]
    cols[t]=[float(re.search(r'median\s+([\d.]+)',l).group(1)) for l in open(f) if 'median' in l and 'sum' not in l]
names=list(cols)
print('layer '+' '.join('%14s'%n for n in names))
for i in range(28):
    print('%5d '%i+' '.join('%14.1f'%cols[n][i] for n in names))
print('sum   '+' '.join('%14.1f'%sum(cols[n]) for n in names))
PY
