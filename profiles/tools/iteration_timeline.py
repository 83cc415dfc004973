"""Kernel timeline of a few iterations of every kind (quiet / validation / penalty update).
   run:    rocprofv3 --kernel-trace --output-format csv -d <dir> -o s -- python profiles/tools/iteration_timeline.py run [workload]
   print:  python profiles/tools/iteration_timeline.py show <dir>/.../s_kernel_trace.csv [first kernel index from the end]"""
import csv, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if sys.argv[1] == "run":
    import torch
    torch.cuda.init()
    import bench
    from dots_socp_amd import meshes
    from dots_socp_amd.socp.solver_socp import AlmSolver
    wl = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "knot"]
    geom, _ = meshes.example(wl["example"], **wl["kw"])
    alm = AlmSolver(wl["n_time"], geom, congestion=wl["congestion"], nit=400, tol=1e-30, time_limit=float("inf"))
    kinds = []
    for i in range(int(os.environ.get("TIMELINE_ITS", "40"))):
        adj = alm.adjust_params.peek_adjust(alm.counter_main + 1)
        val = alm.kkt_validator.will_validate_next()
        kinds.append("P" if adj else ("V" if val else "q"))
        alm.iterate()
    alm.dev.sync()
    print("kinds of the iterations:", "".join(kinds))
    alm.close()
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    rows = rows[-n:]
    t0 = int(rows[0]["Start_Timestamp"])
    prev = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dots::", "")
        print(f"{(s - t0) * 1e-3:9.1f} us  gap {(s - prev) * 1e-3:6.1f}  dur {(e - s) * 1e-3:7.1f}  {name[:60]}")
        prev = e
