"""Merge the mixed-precision keys of a scripts/pmc_bf16.sh traffic.json into the fp32 one of scripts/pmc_conv.sh (both carry the
same per-file source stamps when taken on one build): profiles/pmc_traffic.json then serves bench.py for both precisions.
usage: python scripts/pmc_merge_traffic.py <fp32 traffic.json> <bf16 traffic.json> <out>"""
import json, sys
a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
sa, sb = a.get('stamp', {}).get('kernel_files_sha16', {}), b.get('stamp', {}).get('kernel_files_sha16', {})
for k in ('conv_bf16', 'conv_wgrad9t_bf16'):
    if k in b:
        a[k] = b[k]
for f in ('conv_bf16_halo.hip', 'conv_wgrad_bf16.hip'):      # the stamp of the files the merged keys were taken on
    if f in sb:
        sa[f] = sb[f]
a.setdefault('stamp', {})['kernel_files_sha16'] = sa
a['stamp']['bf16_workload'] = b.get('stamp', {}).get('workload')
json.dump(a, open(sys.argv[3], 'w'), indent=1)
