# The measurement set behind profiles/r02s_* (one MI355X):  bash scripts/r02_final.sh <tag>
set -x
TAG=${1:-r02s}
O=gpurun_out/$TAG; mkdir -p $O
nproc > $O/nproc.txt; lscpu | grep "Model name" >> $O/nproc.txt
timeout 2400 python -m pytest tests -q -s -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_driverargs_line.json 2>/dev/null
timeout 600 python bench.py --force-comm --no-cpu-baseline > $O/bench_forcecomm_line.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline > $O/bench_bf16_b128_line.json 2>/dev/null
timeout 600 python bench.py --dtype bf16 --batch-per-gpu 64 --no-cpu-baseline > $O/bench_bf16_b64_line.json 2>/dev/null
timeout 600 python bench.py --workload audio_tower --no-cpu-baseline > $O/audio_tower_line.json 2>/dev/null
timeout 600 python bench.py --workload vision_tower --no-cpu-baseline > $O/vision_tower_line.json 2>/dev/null
timeout 300 python scripts/dp_overhead.py 64 f32 2>&1 | grep "median\|resident" > $O/dp_overhead.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_f32 -o f32 -- python $R/bench.py --serial --steps 10 --no-cpu-baseline --roofline-steps 0 > $R/$O/prof_f32.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bf16 -o bf16 -- python $R/bench.py --dtype bf16 --batch-per-gpu 128 --serial --steps 10 --no-cpu-baseline --roofline-steps 0 > $R/$O/prof_bf16.log 2>&1
cd $R
bash scripts/pmc_conv.sh $O/pmc > $O/pmc.log 2>&1
bash scripts/pmc_bf16.sh $O/pmc_bf16 128 > $O/pmc_bf16.log 2>&1
timeout 600 python scripts/train_e2e_throughput.py > $O/train_e2e.txt 2>&1
timeout 300 python scripts/embed_throughput.py > $O/embed.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -size +4M -delete
grep "passed\|failed" $O/gpu_tests.log | tail -2
