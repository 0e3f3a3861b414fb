# round 5, first GPU call: the live-gradient bench line (driver arguments) + clocks of both regimes
set -x
O=gpurun_out/r05a; mkdir -p $O
timeout -k 10 900 python bench.py --steps 20 --warmup 5 --quick-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05a/bench_line.json'))
for k in ('value','ms_per_step','loss_first','loss_last','loss_strictly_decreasing','dlogits_nonzero_frac','value_saturated_head','saturated_head'):
    print(k, d.get(k))
print(json.dumps(d.get('secondary'), indent=1)[:3000])
print({k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
