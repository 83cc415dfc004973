"""Device-side timeline of a steady-state window of a rocprofv3 --kernel-trace CSV: busy time (union of kernel
intervals), idle time between kernels, launches -- over the last `n` kernel records before the final 5 %.
usage: python profiles/tools/trace_gaps.py <kernel_trace.csv> [n_kernels]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
end = int(len(rows) * 0.95)
win = rows[max(0, end - n):end]
t0, t1 = int(win[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in win)
busy, cur_e = 0, t0
gaps = []
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_e:
        gaps.append(s - cur_e)
        busy += e - s
        cur_e = e
    elif e > cur_e:
        busy += e - cur_e
        cur_e = e
span = t1 - t0
gaps.sort()
print(f"window: {len(win)} kernels over {span * 1e-6:.3f} ms; device busy {busy * 1e-6:.3f} ms ({100 * busy / span:.1f} %), "
      f"idle {(span - busy) * 1e-6:.3f} ms in {len(gaps)} gaps (median {gaps[len(gaps) // 2] * 1e-3:.2f} us, p90 {gaps[int(len(gaps) * 0.9)] * 1e-3:.2f} us, "
      f"max {gaps[-1] * 1e-3:.1f} us); {span / len(win) * 1e-3:.2f} us per kernel")
