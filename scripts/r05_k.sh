O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -k "split_bf16_engine or algorithm_is_configuration" > $O/tests.log 2>&1; echo rc=$?
grep -a "split-bf16\|issued / direct\|passed\|failed\|Error\|assert" $O/tests.log | cut -c1-400 | head
timeout 300 python bench.py --fp32-conv f2x2_bf16x6 --no-cpu-baseline --no-secondary 2>/dev/null > $O/bench_bx6_line.json
python -c "
import json; d=json.load(open('$O/bench_bx6_line.json')); print(d['value'], d['roofline']['frac'], d['roofline']['peak'], d['roofline']['kernel'][:60], d['roofline'].get('traffic_stale'))"
timeout 300 python bench.py --steps 20 --warmup 5 --quick-cpu-baseline 2>/dev/null > $O/bench_line.json
python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d['roofline']['frac'], d['roofline'].get('traffic_stale'), d['roofline'].get('fp32_alu_occupancy_stale'), d['kernels']['conv_wgrad']['traffic_stale'])"
