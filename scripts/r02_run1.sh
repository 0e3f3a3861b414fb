set -x
mkdir -p gpurun_out/r02a
nproc > gpurun_out/r02a/nproc.txt
timeout 1500 python -m pytest tests/test_layer_parity_gpu.py -q -s -m gpu > gpurun_out/r02a/layer.log 2>&1; echo "layer rc=$?" 
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_boundary.py -q -s -m gpu > gpurun_out/r02a/parity.log 2>&1; echo "parity rc=$?"
timeout 600 python bench.py > gpurun_out/r02a/bench_f32.json 2> gpurun_out/r02a/bench_f32.err; echo "bench rc=$?"
timeout 600 python bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline > gpurun_out/r02a/bench_bf16_b128.json 2> gpurun_out/r02a/bench_bf16.err; echo "bench16 rc=$?"
timeout 600 python bench.py --dtype bf16 --batch-per-gpu 64 --no-cpu-baseline > gpurun_out/r02a/bench_bf16_b64.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02a/prof_bf16 -o bf16 -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --batch-per-gpu 128 --no-cpu-baseline --serial --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/r02a/prof_bf16.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/r02a | head -30; find gpurun_out/r02a -name "*.db" -delete; find gpurun_out/r02a -size +5M -delete
tail -5 gpurun_out/r02a/layer.log; tail -5 gpurun_out/r02a/parity.log
