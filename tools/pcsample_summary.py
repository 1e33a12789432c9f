"""Aggregates rocprofv3's PC-sampling CSV (stochastic method) per instruction: samples, share, issued / stalled split and
stall reasons.  Usage: pcsample_summary.py <rocprofv3 output dir> <output prefix>.  Writes <prefix>_by_pc.csv (one row per
sampled instruction address, sorted by address) -- small enough to travel back from the GPU box -- and prints the totals."""
import collections
import csv
import glob
import os
import sys

raw, prefix = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(os.path.join(raw, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
if not files:
    sys.exit("no pc sampling csv under " + raw)
rows = collections.OrderedDict()
total = 0
cols = None
for f in files:
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        for r in rd:
            total += 1
            key = (r.get("Code_Object_Id", ""), r.get("Code_Object_Offset", r.get("Instruction_Offset", "")), r.get("Instruction", ""), r.get("Instruction_Comment", ""))
            e = rows.setdefault(key, collections.Counter())
            e["n"] += 1
            issued = r.get("Wave_Issued_Instruction", r.get("Wave_Issued", ""))
            if str(issued) in ("1", "True", "true"):
                e["issued"] += 1
            e["type:" + r.get("Instruction_Type", "")] += 1
            e["stall:" + r.get("Stall_Reason", "")] += 1
            m = r.get("Exec_Mask", "")
            if m:
                try:
                    e["lanes"] += bin(int(m, 0) if not m.isdigit() else int(m)).count("1")
                except ValueError:
                    pass
print("columns:", cols)
print("samples:", total, "distinct instructions:", len(rows))
with open(prefix + "_by_pc.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["code_object", "offset", "instruction", "comment", "samples", "issued", "lanes_sum", "types", "stalls"])
    def off(k):
        try:
            return (k[0], int(k[1], 0))
        except ValueError:
            return (k[0], 0)
    for k in sorted(rows, key=off):
        e = rows[k]
        w.writerow(list(k) + [e["n"], e["issued"], e["lanes"],
                              " ".join("%s=%d" % (t[5:], c) for t, c in e.items() if t.startswith("type:")),
                              " ".join("%s=%d" % (t[6:], c) for t, c in e.items() if t.startswith("stall:"))])
agg = collections.Counter()
for e in rows.values():
    for t, c in e.items():
        if t.startswith("stall:") or t.startswith("type:"):
            agg[t] += c
    agg["issued"] += e["issued"]
for t, c in agg.most_common():
    print("%-40s %9d %6.2f %%" % (t, c, 100.0 * c / max(total, 1)))
