"""cProfile of the host side of the iteration loop (knot by default): where the Python / ctypes time of an iteration goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
from dots_socp_amd import meshes
from dots_socp_amd.socp.solver_socp import AlmSolver
geom, _ = meshes.example(sys.argv[1] if len(sys.argv) > 1 else "knot")
alm = AlmSolver(31, geom, nit=1000, tol=1e-30, time_limit=1e9)
for _ in range(5):
    alm.iterate()
alm.dev.sync()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    alm.iterate()
alm.dev.sync()
pr.disable()
print(f"iterations 5..24: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per iteration under cProfile", file=sys.stderr)
pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(28)
for _ in range(155):
    alm.iterate()
alm.dev.sync()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(200):
    alm.iterate()
alm.dev.sync()
pr.disable()
print(f"iterations 180..379: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per iteration under cProfile", file=sys.stderr)
pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
