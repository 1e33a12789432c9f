#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs for the trace kernel: per-dispatch sums."""
import collections, csv, sys
for f in sys.argv[1:]:
    d = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
    for r in csv.DictReader(open(f)):
        if 'rl_trace' in r['Kernel_Name']:
            d[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
            meta[r['Dispatch_Id']] = (r['Grid_Size'], r['VGPR_Count'], r['LDS_Block_Size'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    if not d: continue
    k = sorted(d, key=int)[-1]
    print(f, 'dispatch', k, 'grid/vgpr/lds/ns', meta[k], ' '.join('%s=%.4g' % kv for kv in sorted(d[k].items())))
