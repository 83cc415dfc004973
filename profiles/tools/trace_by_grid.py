"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid size): calls, average / total duration.
usage: python profiles/tools/trace_by_grid.py <kernel_trace.csv> [name filter]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dots::", "")
    if flt and flt not in name:
        continue
    g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    k = (name, g, int(r["Workgroup_Size_X"]))
    acc[k][0] += 1
    acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
tot = sum(v[1] for v in acc.values())
print(f"{'kernel':44s} {'WGs':>7s} {'wg':>5s} {'calls':>7s} {'avg_us':>9s} {'total_ms':>9s} {'%':>6s}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0][:44]:44s} {k[1]:7d} {k[2]:5d} {v[0]:7d} {v[1] / v[0]:9.2f} {v[1] * 1e-3:9.3f} {100 * v[1] / tot:6.2f}")
