"""Registers / spills / LDS of every kernel in a hipcc -save-temps device assembly (.s).  usage: kernel_regs.py file.s [substring]"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', t, re.S):
    name, body = m.group(1), m.group(2)
    try:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dn = name
    if want not in dn:
        continue
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, body) or [None, '-'])[1]
    print('%-72s vgpr %s agpr %s sgpr %s spill %s lds %s' % (dn[:72], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('group_segment_fixed_size')))
