"""Per-kernel averages of rocprofv3 PMC counters over several passes (one counter_collection.csv per pass).
usage: python profiles/tools/sq_summary.py <csv> [<csv> ...]      (prints a table, largest total wave-cycles first)"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dots::", "")
        key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1), int(r["Workgroup_Size"]))
        a = acc[key][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
names = sorted({c for k in acc for c in acc[k]})
rows = []
for key, cs in acc.items():
    avg = {c: cs[c][1] / max(cs[c][0], 1) for c in cs}
    calls = max(v[0] for v in cs.values())
    rows.append((avg.get("SQ_WAVE_CYCLES", 0.0) * calls, key, calls, avg))
rows.sort(reverse=True)
print("kernel | WGs | wg size | calls | " + " | ".join(names) + " | derived")
for _tot, key, calls, avg in rows[:40]:
    d = []
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        d.append(f"wait_any/wave_cycles={avg.get('SQ_WAIT_ANY', 0) / wc:.2f}")
        d.append(f"wait_inst/wave_cycles={avg.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}")
        d.append(f"active/wave_cycles={avg.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}")
    if avg.get("SQ_BUSY_CYCLES"):
        d.append(f"waves_in_flight_avg={wc / avg['SQ_BUSY_CYCLES']:.1f}")
    if avg.get("TCC_REQ_sum"):
        d.append(f"L2_hit={avg.get('TCC_HIT_sum', 0) / max(avg.get('TCC_HIT_sum', 0) + avg.get('TCC_MISS_sum', 0), 1):.2f}")
    print(f"{key[0]} | {key[1]} | {key[2]} | {calls} | " + " | ".join(f"{avg.get(c, float('nan')):.4g}" for c in names) + " | " + " ".join(d))
