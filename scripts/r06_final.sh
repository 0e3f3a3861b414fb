# The measurement set behind profiles/r06*_ (one MI355X):  bash scripts/r06_final.sh <tag> [quick]
# quick: only the serial kernel trace + PMC passes + per-layer traffic (what VERDICT r05 #4 asks about conv1b)
set -x
TAG=${1:-r06v}
QUICK=${2:-}
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
nproc > $O/host.txt; lscpu | grep "Model name" >> $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>/dev/null
if [ -z "$QUICK" ]; then
timeout -k 10 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driverargs_line.json 2>/dev/null
timeout -k 10 600 python bench.py --force-comm --no-cpu-baseline --no-secondary > $O/bench_forcecomm_line.json 2>/dev/null
timeout -k 10 600 python bench.py --fp32-conv f2x2 --no-cpu-baseline --no-secondary > $O/bench_f2x2_line.json 2>/dev/null
timeout -k 10 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline > $O/bench_bf16_b128_line.json 2>/dev/null
timeout -k 10 600 python bench.py --workload audio_tower --no-cpu-baseline > $O/audio_tower_line.json 2>/dev/null
timeout -k 10 600 python bench.py --workload vision_tower --no-cpu-baseline > $O/vision_tower_line.json 2>/dev/null
fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_f32 -o f32 -- python $R/bench.py --serial --steps 10 --no-cpu-baseline --no-secondary --no-saturated --roofline-steps 0 > $R/$O/prof_f32.log 2>&1
if [ -z "$QUICK" ]; then
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bf16 -o bf16 -- python $R/bench.py --serial --steps 10 --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --no-secondary --no-saturated --roofline-steps 0 > $R/$O/prof_bf16.log 2>&1
fi
cd $R
cp $(find $O/prof_f32 -name "*kernel_stats.csv" | head -1) $O/bench_serial_kernel_stats.csv
[ -z "$QUICK" ] && cp $(find $O/prof_bf16 -name "*kernel_stats.csv" | head -1) $O/bench_serial_bf16_b128_kernel_stats.csv
python scripts/wgw_layers.py $(find $O/prof_f32 -name "*kernel_trace.csv" | head -1) conv_w > $O/conv_kernel_durations.txt
python scripts/wino_layers_by_order.py $(find $O/prof_f32 -name "*kernel_trace.csv" | head -1) > $O/wino4_layer_durations.txt
python scripts/wino_layer_fractions.py $O/wino4_layer_durations.txt >> $O/wino4_layer_durations.txt
bash scripts/pmc_conv.sh $O/pmc $O/wino4_layer_durations.txt > $O/pmc.log 2>&1; cp $O/pmc/alu.json $O/pmc_alu.json; cp $O/pmc/traffic.json $O/pmc_traffic.json; cp $O/pmc/summary.txt $O/pmc_summary.txt; cp $O/pmc/traffic_by_layer.txt $O/traffic_by_layer.txt
if [ -z "$QUICK" ]; then
bash scripts/pmc_bf16.sh $O/pmc_bf16 128 > $O/pmc_bf16.log 2>&1; cp $O/pmc_bf16/summary.txt $O/pmc_bf16_summary.txt; python scripts/pmc_merge_traffic.py $O/pmc/traffic.json $O/pmc_bf16/traffic.json $O/pmc_traffic.json
timeout -k 10 600 python scripts/train_e2e_throughput.py concurrent > $O/train_e2e.txt 2>&1
fi
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -size +4M -delete
if [ -z "$QUICK" ]; then
timeout -k 10 2700 python -m pytest tests -q -s -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
grep -a "passed\|failed" $O/gpu_tests.log | tail -2
fi
cat $O/traffic_by_layer.txt
