O=gpurun_out/r05h; mkdir -p $O
for A in f4x4 f2x2_bf16x6 f4x4 f2x2_bf16x6; do
timeout 300 python bench.py --fp32-conv $A --no-cpu-baseline --no-secondary --no-saturated --steps 30 --warmup 8 --roofline-steps 3 2>/dev/null > $O/line_$A.json
python - <<PY
import json
d=json.load(open('$O/line_$A.json'))
print('$A', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})
PY
done
