"""Per-kernel ratios from a scripts/pmc_summarize.py summary (matrix pipe busy, park / issue-stall shares, LDS use).
usage: python scripts/pmc_ratios.py <summary.txt> [substring ...]"""
import re, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2:] or ['']
for b in re.split(r'\n== ', '\n' + txt):
    name = b.split('\n')[0].replace('== ', '')
    if not any(w in name for w in want):
        continue
    d, n = {}, 0
    for line in b.split('\n')[1:]:
        m = re.match(r'\s+(\S+)\s+mean/dispatch\s+([\d.]+)\s+dispatches\s+(\d+)', line)
        if m:
            d[m.group(1)] = float(m.group(2)); n = int(m.group(3))
    if 'GRBM_GUI_ACTIVE' not in d or not d.get('SQ_INSTS_MFMA'):
        continue
    cyc = d['GRBM_GUI_ACTIVE'] / 8          # summed over the 8 XCDs
    wc = d['SQ_WAVE_CYCLES']
    print('%-46s n=%3d kcyc %6.0f mfma_busy %.3f | wave: parked %.2f issue-stall %.2f issuing %.2f | lds_busy %.2f | per mfma: valu %.2f salu %.2f lds %.2f vmem %.3f'
          % (name[:46], n, cyc / 1e3, d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc), d['SQ_WAIT_ANY'] / wc, d['SQ_WAIT_INST_ANY'] / wc,
             d['SQ_ACTIVE_INST_ANY'] / wc, d['SQ_LDS_IDX_ACTIVE'] / (256 * cyc), d['SQ_INSTS_VALU'] / d['SQ_INSTS_MFMA'],
             d['SQ_INSTS_SALU'] / d['SQ_INSTS_MFMA'], d['SQ_INSTS_LDS'] / d['SQ_INSTS_MFMA'], (d['SQ_INSTS_VMEM_RD'] + d['SQ_INSTS_VMEM_WR']) / d['SQ_INSTS_MFMA']))
