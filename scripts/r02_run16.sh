O=gpurun_out/r02p; mkdir -p $O
L3_TWO_STREAMS=0 L3_PROFILE_VERBOSE=1 timeout 300 python scripts/step_profile.py 128 cnn_L3_melspec2 3 bf16 > $O/sp_bf16.log 2>&1; grep "ms/step" $O/sp_bf16.log; grep "fam=2 conv2d_1 \|fam=2 conv2d_8 " $O/sp_bf16.log | tail -4
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
L3_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/scripts/step_profile.py 128 cnn_L3_melspec2 3 bf16 > /dev/null 2>&1
cd $R; grep "first" $O/prof/p_kernel_stats.csv | cut -c1-200; find $O -name "*trace.csv" -delete
