#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs for the trace kernel: per-dispatch sums of the last
rl_trace_kernel dispatch of each pass.

    python tools/pmc_summary.py <pass.csv>...                            # one line per pass (human readable)
    python tools/pmc_summary.py --json OUT.json --bench BENCH.json <pass.csv>...
        merges the passes into OUT.json = {build_id, config, fetch, rays_per_launch, kernel_ns, counters{...}}: what
        bench.py's roofline.executed reads.  BENCH.json is the JSON line bench.py printed under one of the passes (same
        command, hence the same rays per launch and the same build id)."""
import collections
import csv
import json
import sys

args = sys.argv[1:]
out_json = bench_json = None
while args and args[0] in ("--json", "--bench"):
    if args[0] == "--json":
        out_json = args[1]
    else:
        bench_json = args[1]
    args = args[2:]
merged, kernel_ns = {}, []
for f in args:
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(f)):
        if 'rl_trace' in r['Kernel_Name']:
            d[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
            meta[r['Dispatch_Id']] = (r['Grid_Size'], r['VGPR_Count'], r['LDS_Block_Size'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    if not d:
        continue
    k = sorted(d, key=int)[-1]
    print(f, 'dispatch', k, 'grid/vgpr/lds/ns', meta[k], ' '.join('%s=%.4g' % kv for kv in sorted(d[k].items())))
    merged.update(d[k])
    kernel_ns.append(meta[k][3])
if out_json:
    b = json.loads(open(bench_json).read().strip().splitlines()[-1])
    out = {"build_id": b["config"]["build_id"], "config": b["config"]["config"],
           "fetch": "lds" if "in LDS" in b["config"]["workload"] else "global",
           "rays_per_launch": b["roofline"]["rays_per_launch"], "paths_per_launch": b["config"]["paths_per_launch"],
           "kernel_ns": sum(kernel_ns) / len(kernel_ns), "counters": merged,
           "note": "sums over all XCDs / SEs of the last rl_trace_kernel dispatch of each rocprofv3 --pmc pass "
                   "(tools/profile_round.sh); GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_*_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles"}
    json.dump(out, open(out_json, "w"), indent=1, sort_keys=True)
    print("wrote", out_json)
