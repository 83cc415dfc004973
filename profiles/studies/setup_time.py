import sys, time
sys.path.insert(0,'.')
import numpy as np, scipy.sparse as sp
from dots_socp_amd import geometry, meshes, frontal
from dots_socp_amd.device import DeviceProblem
for name,kw in (("sphere",dict(level=5)),("torus",dict(nu=400,nv=250))):
    g,_=meshes.example(name,**kw)
    t=time.perf_counter(); plan=geometry.build_plan(31,g,reorder="nd"); t1=time.perf_counter()-t
    t=time.perf_counter(); dev=DeviceProblem(31,g,lap_solver="modal_pcg",plan=plan); t2=time.perf_counter()-t
    V=plan.n_vertices
    K=sp.csr_matrix((plan.lap_val,plan.lap_col,plan.lap_rowptr),shape=(V,V))
    t=time.perf_counter(); ff=frontal.factorize(K,plan.mass_vert,plan.time_eigs,plan.dissection,pitch=32,numeric=False); t3=time.perf_counter()-t
    t=time.perf_counter(); dev.setup_frontal(); t4=time.perf_counter()-t
    print(name, "build_plan %.3f  dots_create %.3f  symbolic(py) %.3f  setup_frontal total %.3f (device numeric+upload = %.3f)"%(t1,t2,t3,t4,t4-t3))
    dev.close()
