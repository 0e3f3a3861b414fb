"""Instruction mix of the MFMA-carrying basic blocks of one kernel in a hipcc -S listing:
   python scripts/probes/isa_blocks.py file.s <kernel-name-substring> [min_mfma]"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\S+):', l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1); continue
    t = l.strip().split()
    if t and not t[0].startswith(('.', ';')):
        cur.append(t[0])
blocks.append((name, cur))
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_pk_'): return 'valu_pk'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'scratch_')): return 'vmem:' + op.split('_')[0]
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_barrier'): return 's_barrier'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith('s_'): return 'salu'
    return op
for name, ops in blocks:
    h = collections.Counter(cls(o) for o in ops)
    if h['mfma'] >= min_mfma:
        print(name, len(ops), dict(sorted(h.items())))
        v = collections.Counter(o for o in ops if o.startswith('v_') and not o.startswith('v_mfma'))
        print('   ', dict(v.most_common(12)))
