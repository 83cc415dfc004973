"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes: both do not fit one pass).

usage: python profiles/tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
           [--calib KERNEL_SUBSTRING READ_BYTES WRITE_BYTES]

The counters' absolute scale on gfx950 depends on the access width (the guide: FETCH_SIZE reports half the bytes
of 16-B-per-lane streaming reads, other widths uncalibrated), so the run is calibrated on a kernel of the same
trace whose byte count is known exactly (k_scale / k_divide: one 8-B load and one 8-B store per element, the
access width of every kernel of this library): factor = known bytes / reported value.  Output: per kernel the
number of launches, the average reported counters and the calibrated bytes per launch."""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dots::", "")
        key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        acc[key][0] += 1
        acc[key][1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    calib = None
    if "--calib" in sys.argv:
        i = sys.argv.index("--calib")
        calib = (sys.argv[i + 1], float(sys.argv[i + 2]), float(sys.argv[i + 3]))
    F, W = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    kf = kw = None
    if calib:
        for (name, grid), (n, tot) in F.items():
            if calib[0] in name and tot > 0:
                cand = calib[1] / (tot / n)
                kf = cand if kf is None else kf   # first matching grid (the large streaming launch is listed first by size below)
        for (name, grid), (n, tot) in W.items():
            if calib[0] in name and tot > 0:
                kw = calib[2] / (tot / n) if kw is None else kw
    res = {"calibration": {"kernel": calib[0] if calib else None, "fetch_factor_bytes_per_unit": kf, "write_factor_bytes_per_unit": kw},
           "kernels": []}
    for key in sorted(set(F) | set(W), key=lambda k: -(F.get(k, [0, 0])[1])):
        n = F.get(key, W.get(key))[0]
        f = F.get(key, [1, 0.0])
        w = W.get(key, [1, 0.0])
        e = {"kernel": key[0], "workgroups": key[1], "launches": n, "fetch_size_avg": f[1] / max(f[0], 1), "write_size_avg": w[1] / max(w[0], 1)}
        if kf and kw:
            e["read_bytes_per_launch"] = kf * e["fetch_size_avg"]
            e["written_bytes_per_launch"] = kw * e["write_size_avg"]
        res["kernels"].append(e)
    if "--front-levels" in sys.argv and kf and kw:
        levels = int(sys.argv[sys.argv.index("--front-levels") + 1])
        rd = sum(e["read_bytes_per_launch"] * e["launches"] for e in res["kernels"] if e["kernel"].startswith("k_front_"))
        wr = sum(e["written_bytes_per_launch"] * e["launches"] for e in res["kernels"] if e["kernel"].startswith("k_front_"))
        n_fwd = sum(e["launches"] for e in res["kernels"] if e["kernel"].startswith("k_front_fwd"))
        solves = n_fwd / levels
        res["front_solve"] = {"solves": solves, "read_bytes_per_solve": rd / solves, "written_bytes_per_solve": wr / solves,
                              "bytes_per_solve": (rd + wr) / solves}
        print(res["front_solve"])
    json.dump(res, open(out, "w"), indent=1)
    for e in res["kernels"][:40]:
        print(e)


if __name__ == "__main__":
    main()
