set -x
O=gpurun_out/r02b; mkdir -p $O
timeout 2400 python -m pytest tests -q -s -m gpu -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
timeout 900 python bench.py > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench rc=$?"
timeout 600 python bench.py --force-comm --no-cpu-baseline > $O/bench_f32_forcecomm.json 2> $O/bench_fc.err; echo "bench fc rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_f32_driverargs.json 2>/dev/null
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_f32 -o f32 -- python $R/bench.py --serial --steps 10 --no-cpu-baseline --roofline-steps 0 > $R/$O/prof_f32.log 2>&1
cd $R
OUT=$O/pmc bash scripts/pmc_conv.sh $O/pmc > $O/pmc.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -size +4M -delete
tail -3 $O/gpu_tests.log
