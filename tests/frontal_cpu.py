"""numpy restatement of the two sweeps of the device's multifrontal solve (csrc/kernels_front.hip), reading the
same flat arrays (dots_socp_amd/frontal.py:FrontalFactor).  TEST INFRASTRUCTURE: validates the host-side
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


# ------------------------------------------------------------------------------------------------
# merged tree heights (the device's form of the sweeps when dots_front_desc.band_ptr is given):
# the nodes of a band of heights [lo, hi) that hang together become ONE node whose block is
#     F' = [ L'^-1 ; G' ],  L'^-1 block lower triangular over the members (children first), G' = the update operator
# of the band's top node on ITS boundary, both computed from the members' own blocks (no new factorisation):
#     member s, child c inside the band:   row block s, columns of c's subtree = -L_s^-1 U_c[rows of sep_s]
#                                          U_s[columns of c's subtree]         = U_c[rows of bd_s] - G_s U_c[rows of sep_s]
#     own columns:                         L_s^-1 and U_s = G_s
# ------------------------------------------------------------------------------------------------
def heights(ff):
    nn = ff.node_n.size
    h = np.zeros(nn, dtype=np.int64)
    for lv in range(ff.level_ptr.size - 1):
        h[ff.level_nodes[ff.level_ptr[lv]:ff.level_ptr[lv + 1]]] = lv
    return h


def merge(ff, cuts):
    """cuts: increasing heights starting with 0 and ending with the number of levels.  Returns the list of bands,
    each a list of merged nodes (dicts: sep vertices, boundary vertices, F' (m', n', A), children [(node, rows)])."""
    A = ff.n_modes
    nn = ff.node_n.size
    h = heights(ff)
    parent = np.full(nn, -1, dtype=np.int64)
    for p in range(nn):
        for c in ff.node_child[p]:
            if c >= 0:
                parent[c] = p
    band_of = np.searchsorted(np.asarray(cuts), h, side="right") - 1
    F = ff.values[:, :A]
    pulls = (ff.pull0, ff.pull1)
    root = np.arange(nn)
    for p in range(nn - 1, -1, -1):
        if parent[p] >= 0 and band_of[parent[p]] == band_of[p]:
            root[p] = root[parent[p]]
    bands = [[] for _ in range(len(cuts) - 1)]
    for p in range(nn):
        if root[p] != p:
            continue
        members = np.flatnonzero(root == p)                 # ascending = children first
        n_of = ff.node_n[members].astype(np.int64)
        off = dict(zip(members.tolist(), (np.cumsum(n_of) - n_of).tolist()))
        npr = int(n_of.sum())
        bp = int(ff.node_b[p])
        iop = int(ff.node_ioff[p])
        sepv = np.concatenate([ff.front_idx[ff.node_ioff[s]:ff.node_ioff[s] + ff.node_n[s]] for s in members]) if npr else np.zeros(0, np.int32)
        bdv = ff.front_idx[iop + ff.node_n[p]:iop + ff.node_n[p] + bp]
        rowpos = {int(v): i for i, v in enumerate(np.concatenate([sepv, bdv]))}
        Fm = np.zeros((npr + bp, npr, A))
        U, c0 = {}, {}
        ext = []
        for s in members.tolist():
            n, b = int(ff.node_n[s]), int(ff.node_b[s])
            io, fo = int(ff.node_ioff[s]), int(ff.node_foff[s])
            Fs = F[fo:fo + (n + b) * n].reshape(n + b, n, A)
            o = off[s]
            c0[s] = o
            kids = [(k, int(c)) for k, c in enumerate(ff.node_child[s]) if c >= 0]
            for k, c in kids:
                if root[c] == p:
                    c0[s] = min(c0[s], c0[c])
            Us = np.zeros((b, o + n - c0[s], A))
            Fm[o:o + n, o:o + n] = Fs[:n]
            Us[:, o - c0[s]:] = Fs[n:]
            for k, c in kids:
                pl = pulls[k][io:io + n + b]
                if root[c] != p:      # a node of a lower band: its update rows land on these rows of the merged front
                    rows = np.full(int(ff.node_b[c]), -1, dtype=np.int64)
                    for fpos in np.flatnonzero(pl >= 0):
                        rows[pl[fpos]] = rowpos[int(ff.front_idx[io + fpos])]
                    assert (rows >= 0).all()
                    ext.append((c, rows))
                    continue
                Uc = U.pop(c)
                wc = Uc.shape[1]
                Ucs = np.zeros((n + b, wc, A))
                has = pl >= 0
                Ucs[has] = Uc[pl[has]]
                acc = np.einsum("ika,kja->ija", Fs, Ucs[:n])
                Fm[o:o + n, c0[c]:c0[c] + wc] = -acc[:n]
                Us[:, c0[c] - c0[s]:c0[c] - c0[s] + wc] = Ucs[n:] - acc[n:]
            U[s] = Us
        Up = U.pop(p)
        assert not U and Up.shape[1] == npr
        Fm[npr:] = Up
        bands[band_of[p]].append(dict(node=p, sep=sepv, bd=bdv, F=Fm, children=ext))
    return bands


def solve_merged(ff, bands, rhs, top_inverse=False):
    """``top_inverse``: the nodes of the top band (no boundary rows) hold S^-1 = L'^-T L'^-1 instead of L'^-1: their forward
    step already gives x, the backward sweep starts one band lower (dots_front_desc.top_inverse)."""
    A = ff.n_modes
    y = np.zeros((ff.n_vertices, A))
    x = np.zeros((ff.n_vertices, A))
    u = {}
    for k, band in enumerate(bands):
        for g in band:
            n, b = g["sep"].size, g["bd"].size
            acc = np.zeros((n + b, A))
            for c, rows in g["children"]:
                acc[rows] += u.pop(c)
            if top_inverse and k == len(bands) - 1:
                assert b == 0
                Sinv = np.einsum("kia,kja->ija", g["F"], g["F"])
                x[g["sep"]] = np.einsum("ija,ja->ia", Sinv, rhs[g["sep"]] - acc)
                continue
            out = np.einsum("ija,ja->ia", g["F"], rhs[g["sep"]] - acc[:n])
            y[g["sep"]] = out[:n]
            if b:
                u[g["node"]] = acc[n:] + out[n:]
    assert not u
    for band in (bands[:-1] if top_inverse else bands)[::-1]:
        for g in band:
            v = np.concatenate([y[g["sep"]], -x[g["bd"]]], axis=0)
            x[g["sep"]] = np.einsum("jia,ja->ia", g["F"], v)
    return x
