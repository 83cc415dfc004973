"""numpy restatement of the two sweeps of the device's multifrontal solve (csrc/kernels_front.hip), reading the
same flat arrays (dots-socp_amd/frontal.py:FrontalFactor).  TEST INFRASTRUCTURE: validates the host-side
factorisation against scipy on CPU and stands in for the device in the gloo tests; the product has no CPU path."""
import numpy as np


def solve(ff, rhs):
    """rhs: (V, n_modes) -> x with (K + shift_a M) x[:, a] = rhs[:, a]."""
    A = ff.n_modes
    nn = ff.node_n.size
    y = np.zeros((ff.n_vertices, A))
    x = np.zeros((ff.n_vertices, A))
    u = np.zeros((max(ff.update_rows, 1), A))
    F = ff.values[:, :A]
    for lv in range(ff.level_ptr.size - 1):
        for p in ff.level_nodes[ff.level_ptr[lv]:ff.level_ptr[lv + 1]]:
            n, b = int(ff.node_n[p]), int(ff.node_b[p])
            io, fo, uo = int(ff.node_ioff[p]), int(ff.node_foff[p]), int(ff.node_uoff[p])
            idx = ff.front_idx[io:io + n + b]
            acc = np.zeros((n + b, A))
            for k, pull in enumerate((ff.pull0, ff.pull1)):
                c = ff.node_child[p, k]
                if c < 0:
                    continue
                pl = pull[io:io + n + b]
                has = pl >= 0
                acc[has] += u[ff.node_uoff[c] + pl[has]]
            w = rhs[idx[:n]] - acc[:n]
            Fp = F[fo:fo + (n + b) * n].reshape(n + b, n, A)
            out = np.einsum("ija,ja->ia", Fp, w)
            y[idx[:n]] = out[:n]
            if b:
                u[uo:uo + b] = acc[n:] + out[n:]
    for lv in range(ff.level_ptr.size - 2, -1, -1):
        for p in ff.level_nodes[ff.level_ptr[lv]:ff.level_ptr[lv + 1]]:
            n, b = int(ff.node_n[p]), int(ff.node_b[p])
            io, fo = int(ff.node_ioff[p]), int(ff.node_foff[p])
            idx = ff.front_idx[io:io + n + b]
            Fp = F[fo:fo + (n + b) * n].reshape(n + b, n, A)
            v = np.concatenate([y[idx[:n]], -x[idx[n:]]], axis=0)
            x[idx[:n]] = np.einsum("jia,ja->ia", Fp, v)
    return x
