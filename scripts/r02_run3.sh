set -x
O=gpurun_out/r02c; mkdir -p $O
timeout 1500 python -m pytest tests/test_layer_parity_gpu.py -q -s -m gpu > $O/layer.log 2>&1; echo "layer rc=$?"
timeout 1800 python -m pytest tests/test_parity_gpu.py -q -s -m gpu > $O/parity.log 2>&1; echo "parity rc=$?"
timeout 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline > $O/bench_bf16_b128.json 2> $O/bench_bf16.err; echo "bench16 rc=$?"
timeout 600 python bench.py --dtype bf16 --batch-per-gpu 64 --no-cpu-baseline > $O/bench_bf16_b64.json 2>/dev/null
L3_BF16_CONV_OUT=0 timeout 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline > $O/bench_bf16_b128_noout.json 2>/dev/null
timeout 600 python bench.py --force-comm --no-cpu-baseline > $O/bench_f32_forcecomm.json 2> $O/bench_fc.err; echo "bench fc rc=$?"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bf16 -o bf16 -- python $R/bench.py --dtype bf16 --batch-per-gpu 128 --serial --steps 10 --no-cpu-baseline --roofline-steps 0 > $R/$O/prof_bf16.log 2>&1
cd $R
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -size +4M -delete
tail -3 $O/layer.log; tail -3 $O/parity.log
