# baseline of the restored checkpoint: GPU tests, bf16 b128 bench + per-launch listing + PMC of the bf16 kernels, default bench
set -x
O=gpurun_out/r05r; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --no-secondary > $O/bench_bf16_b128_line.json 2>$O/bench_bf16.err
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_line.json 2>$O/bench.err
cd /tmp && export TMPDIR=/tmp
L3_TWO_STREAMS=0 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr -o bf16 -- python $R/scripts/step_profile.py 128 cnn_L3_melspec2 6 bf16 > $R/$O/tr.log 2>&1
cd $R
python scripts/kernels_in_order.py $(find $O/tr -name "*kernel_trace.csv" | head -1) > $O/bf16_b128_kernels_in_order.txt
find $O/tr -name "*.db" -delete; find $O/tr -name "*kernel_trace.csv" -delete
bash scripts/pmc_bf16.sh $O/pmc 128 > $O/pmc.log 2>&1
timeout -k 10 2400 python -m pytest tests -q -s -m gpu -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
grep -a "passed\|failed" $O/gpu_tests.log | tail -2
