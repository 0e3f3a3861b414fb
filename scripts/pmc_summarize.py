"""Aggregates rocprofv3 counter_collection CSVs per kernel name (mean per dispatch)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, '*', '*counter_collection.csv')) + glob.glob(os.path.join(root, '*', '*', '*counter_collection.csv')):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get('Kernel_Name', '')
            short = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void l3::', '').replace('l3::', '')[:48]
            c = r.get('Counter_Name')
            v = float(r.get('Counter_Value', 0))
            a = agg[short][c]
            a[0] += v
            a[1] += 1
keys = ['conv_wgrad_wino_kernel', 'conv_bf16_halo_kernel', 'conv_igemm_bf16in_kernel', 'conv_wgrad9t_kernel', 'conv_wino_kernel', '(anonymous namespace)::conv_wino', 'conv_igemm_glds_kernel', 'conv_igemm_kernel', 'conv_wgrad9_kernel', 'conv_wgrad_kernel', 'conv_dgrad_small']
for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
    if not any(k.startswith(x[:20]) for x in keys) and 'bn_' not in k and 'wino' not in k and 'halo' not in k and 'bf16' not in k:
        continue
    print('==', k)
    for c, (s, n) in sorted(agg[k].items()):
        print('   %-28s mean/dispatch %16.1f   dispatches %d' % (c, s / n, n))

# fp32 MFMA kernels: share of the SIMDs' cycles the matrix pipe is busy (64 cycles per v_mfma_f32_32x32x2_f32), and the same with the
# kernel's other VALU instructions at the ~4.3 cycles of matrix time each of them takes on gfx950 (scripts/mfma_mix.hip):
# the second number is how full the SIMDs' fp32 ALUs are.  SQ_WAVE_CYCLES counts quad-cycles per wave; 1024 SIMDs.
# (conv_wgrad_wino_kernel's loop arithmetic is packed since round 4 -- v_pk_fma_f32, ~5.6 cycles each beside the fp32 MFMA,
# scripts/mfma_mix_bf16.hip -- and is priced at that.)
print()
print('# fp32 ALU occupancy model (matrix pipe busy | + 4.3 cycles per other VALU instruction), of all SIMD cycles of the launch')
alu = {'method': 'rocprofv3 --pmc (SQ_INSTS_MFMA, SQ_INSTS_VALU, SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAVES), mean per launch; '
                 'matrix_pipe_busy = MFMA busy cycles / (1024 SIMDs x kernel cycles); alu_busy adds the other VALU instructions at 4.3 '
                 'matrix cycles each (scripts/mfma_mix.hip; 5.6 for the packed instructions of conv_wgrad_wino_kernel)'}
for k in sorted(agg):
    c = agg[k]
    if not all(x in c for x in ('SQ_INSTS_MFMA', 'SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAVES', 'SQ_VALU_MFMA_BUSY_CYCLES')):
        continue
    mean = lambda x: c[x][0] / c[x][1]
    mf, va = mean('SQ_INSTS_MFMA'), mean('SQ_INSTS_VALU') - mean('SQ_INSTS_MFMA')
    if mf < 1e5 or ('wino' not in k and 'wgrad' not in k and 'igemm' not in k):
        continue
    simd_cycles = mean('SQ_WAVE_CYCLES') * 4.0 / mean('SQ_WAVES') * 1024.0
    busy = mean('SQ_VALU_MFMA_BUSY_CYCLES')
    vcost = 5.6 if 'conv_wgrad_wino_kernel' in k else 4.3
    print('   %-34s MFMA %7.2f M  other VALU %7.2f M (%.2f per MFMA)  matrix pipe %.3f | with VALU %.3f' % (
        k, mf / 1e6, va / 1e6, va / mf, busy / simd_cycles, (busy + vcost * va) / simd_cycles))
    alu[k] = {'mfma_per_launch': mf, 'other_valu_per_launch': va, 'valu_per_mfma': va / mf, 'matrix_pipe_busy': busy / simd_cycles,
              'alu_busy': (busy + vcost * va) / simd_cycles, 'cycles_per_valu': vcost}
print()
import json as _json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from l3embedding_amd import _build as _b
# staleness key: the kernels these counters were taken on; bench.py prints "traffic_stale": true when the tree differs
STAMP = {'csrc_sha16': _b.source_hash(), 'kernel_files_sha16': {f: _b.source_hash([f]) for f in
         ('conv_wino4.hip', 'conv_wgrad_wino.hip', 'conv_bf16_halo.hip', 'conv_wgrad_bf16.hip') if os.path.exists(os.path.join(_b.CSRC, f))},
         'workload': os.environ.get('PMC_WORKLOAD', 'scripts/step_profile.py 64 cnn_L3_melspec2 1 (live head)')}
alu['stamp'] = STAMP
with open(os.path.join(root, 'alu.json'), 'w') as fh:
    _json.dump(alu, fh, indent=1)

# machine-readable HBM traffic of the dominant kernel, corrected as MI355X_MICROARCH.md "HBM"
# prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of
# wide (16 B/lane) coalesced reads -> x2; WRITE_SIZE is taken as reported.
import json


def traffic_of(pred, label):
    dom = [k for k in agg if pred(k)]
    f = w = n = 0.0
    for k in dom:
        if 'FETCH_SIZE' in agg[k] and 'WRITE_SIZE' in agg[k]:
            f += agg[k]['FETCH_SIZE'][0]
            w += agg[k]['WRITE_SIZE'][0]
            n += agg[k]['FETCH_SIZE'][1]
    if not n:
        return None
    return {'kernel': ', '.join(sorted(dom)) + label, 'launches_sampled': int(n),
            'fetch_bytes_per_launch': f / n * 1024 * 2, 'write_bytes_per_launch': w / n * 1024,
            'hbm_bytes_per_launch': f / n * 1024 * 2 + w / n * 1024}


out = {'method': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB units; '
                 'FETCH_SIZE x2 (gfx950 wide-load correction, MI355X_MICROARCH.md HBM section); mean per launch'}
for key, pred, label in (('conv_wino4', lambda k: 'conv_wino4_kernel' in k, ' (F(4x4,3x3) forward + dgrad launches)'),
                         ('conv_wino', lambda k: 'conv_wino_kernel' in k, ' (F(2x2,3x3) forward + dgrad launches)'),
                         ('conv_wgrad9t', lambda k: k.startswith('conv_wgrad9t_kernel'), ' (weight-gradient launches)'),
                         ('conv_wgrad_wino', lambda k: k.startswith('conv_wgrad_wino_kernel'), ' (Winograd weight-gradient launches)'),
                         ('conv_bf16', lambda k: k.startswith('conv_igemm_bf16') or k.startswith('conv_bf16_halo'), ' (bf16 forward + dgrad launches)'),
                         ('conv_wgrad9t_bf16', lambda k: k.startswith('conv_wgrad_bf16_tr'), ' (bf16 weight-gradient launches)')):
    t = traffic_of(pred, label)
    if t:
        out[key] = t
# the elementwise family (BatchNorm / ReLU / pool kernels and their reduction finals) of ONE step: counter bytes per step, for
# bench.py's `elementwise.traffic` (the profiled run is `step_profile.py <B> <model> 1`: 2 warm-up + 1 timed + 1 profiled = 4 steps)
EW = ('bn_', 'fast_final', 'fast_prefinal', 'global_maxpool', 'maxpool', 'relu_', 'colreduce', 'colfinal', 'bn_moving')
ew_f = ew_w = 0.0
ew_n = 0
for k in agg:
    if k.startswith(EW) and 'FETCH_SIZE' in agg[k] and 'WRITE_SIZE' in agg[k]:
        ew_f += agg[k]['FETCH_SIZE'][0]
        ew_w += agg[k]['WRITE_SIZE'][0]
        ew_n += agg[k]['FETCH_SIZE'][1]
steps_profiled = int(os.environ.get('PMC_STEPS', '4'))
if ew_n:
    out['elementwise'] = {'kernel': 'bn_* / fast_final / pool / relu / colreduce kernels (the engine\'s `elementwise` family)',
                          'launches_sampled': int(ew_n), 'steps_sampled': steps_profiled,
                          'fetch_bytes_per_step': ew_f * 1024 * 2 / steps_profiled, 'write_bytes_per_step': ew_w * 1024 / steps_profiled,
                          'hbm_bytes_per_step': (ew_f * 1024 * 2 + ew_w * 1024) / steps_profiled}
out['stamp'] = STAMP
with open(os.path.join(root, 'traffic.json'), 'w') as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out))
