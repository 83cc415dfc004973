// Direct solve of the modal surface problems  (K + (sigma_a + eps) M) x_a = b_a  for all time modes a:
// the two triangular sweeps of a multifrontal Cholesky factorisation (replaces the per-iteration
// SuperLU solves of the reference, utils/laplacian_inverse_socp.py:46-60; the factor itself is built once
// per solve by dots-socp_amd/frontal.py, as the reference builds its T+1 LU factors at :40-44).
//
// One nested-dissection tree is shared by all modes.  Node p eliminates n_p separator vertices and
// touches b_p boundary vertices of its ancestors; its dense block per mode is
//     F_p = [ L_pp^-1 ; G_p ],   G_p = A_bs A_ss^-1,   (n_p + b_p) x n_p,   stored [row][col][mode]
// so that BOTH sweeps are batched dense matrix-vector products that stream F once, with the mode index
// fastest (a wavefront reads two 256-byte runs per load at T = 31, like every other kernel of the path):
//     forward   w   = b[sep_p] - (update rows pulled from the two children)
//               y_p = L_pp^-1 w                    rows [0, n_p) of F_p (lower triangle only)
//               u_p = (children's updates on bd_p) + G_p w          rows [n_p, n_p + b_p)
//     backward  x_p = F_p^T [ y_p ; -x[bd_p] ]
// Every child owns one plane of its parent's update buffer W (front-ordered, position map `cmap` precomputed from
// the pull maps of the ABI): the child writes its update rows there, the parent reads its two planes contiguously.
// No two writers share an address, so no atomics are needed and results are deterministic; entries no child
// writes stay zero.  All nodes of one tree height are independent: one launch per height and sweep, a
// workgroup = (node, block of rows | columns), the dot products split over the workgroup's lanes that do not
// index the mode and folded by wave shuffles + LDS.
// HBM-bound: one solve reads sum_p (n_p (n_p + 1) / 2 + b_p n_p) * modes * 8 bytes twice and touches the
// vectors (V * modes * 8 bytes) a handful of times.
#include "dots_dev.h"

#include <algorithm>
#include <vector>

namespace dots {

struct FrontArgs {
    int sh, TP, ncol;          // mode pitch (log2, value) and live modes
};

__device__ __forceinline__ int64_t front_row(const FrontDev &f, int k) { return f.vmap ? f.vmap[k] : k; }

// VEC consecutive modes per lane (VEC = 2: 16-byte loads, twice as many parts q of a dot product per workgroup).
template <int VEC> struct Vd { double v[VEC]; };
template <int VEC> __device__ __forceinline__ Vd<VEC> vload(const double *p) {
    Vd<VEC> o;
    if (VEC == 2) {
        const double2 t = *reinterpret_cast<const double2 *>(p);
        o.v[0] = t.x;
        o.v[VEC - 1] = t.y;
    } else {
        o.v[0] = *p;
    }
    return o;
}
template <int VEC> __device__ __forceinline__ void vstore(double *p, const Vd<VEC> &x) {
    if (VEC == 2) *reinterpret_cast<double2 *>(p) = make_double2(x.v[0], x.v[VEC - 1]);
    else *p = x.v[0];
}

// Sum `acc` over the threads of the workgroup that share their modes (the parts q of the dot products): lanes of a
// wavefront first (xor-shuffles), then one LDS slot per (row, wavefront, mode); the caller reads them back with
// front_folded after the barrier inside.  TPv = TP / VEC lanes hold one row part; TPv <= 64.
template <int NB, int RB, int VEC>
__device__ __forceinline__ void front_fold(Vd<VEC> (&acc)[RB], double *red, int TP, int a, int tid) {
    constexpr int NW = NB / 64;
    const int TPv = TP / VEC;
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            double s = acc[r].v[c];
            for (int o = 32; o >= TPv; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[r].v[c] = s;
        }
    if ((tid & 63) < TPv) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < VEC; ++c) red[(r * NW + (tid >> 6)) * TP + a + c] = acc[r].v[c];
    }
    __syncthreads();
}
template <int NB, int VEC>
__device__ __forceinline__ Vd<VEC> front_folded(const double *red, int r, int TP, int a) {
    constexpr int NW = NB / 64;
    Vd<VEC> s;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += red[(r * NW + w) * TP + a + c];
        s.v[c] = t;
    }
    return s;
}
// one thread per mode and row part when a row of modes is wider than a wavefront (TP / VEC > 64)
template <int NB, int RB>
__device__ __forceinline__ void front_fold_wide(Vd<1> (&acc)[RB], double *red, int tid) {
#pragma unroll
    for (int r = 0; r < RB; ++r) red[r * NB + tid] = acc[r].v[0];
    __syncthreads();
}
template <int NB>
__device__ __forceinline__ Vd<1> front_folded_wide(const double *red, int r, int sh, int a) {
    Vd<1> s;
    s.v[0] = 0.0;
    for (int k = 0; k < (NB >> sh); ++k) s.v[0] += red[r * NB + (k << sh) + a];
    return s;
}

// forward sweep of one tree height.  Workgroup = (node, rb <= RB rows); thread = (VEC modes from a, part q of the dot
// product).  Every load of the loop body is unconditional (rows past the block are clamped to its first row and
// their sums dropped; leaves read their "children's" planes from the zero pad at the start of W; the upper triangle
// of L^-1 is stored as zeros), so that the compiler issues the RB + 3 loads of a step back to back and waits once.
// RB is the level's exact block size (1, 2 or 4: no duplicate loads); LEAF (tree height 0): no update planes to read.
template <int NB, int RB, bool VMAP, bool LEAF, int VEC>
__global__ __launch_bounds__(NB) void k_front_fwd(FrontArgs g, FrontDev f, const FrontWork *__restrict__ desc, int rb, const double *__restrict__ bhat,
                                                  double *__restrict__ Y) {
    __shared__ double red[RB * (NB / 64) * 64 * VEC];
    const FrontWork wk = desc[blockIdx.x];
    const FrontNode &nd = wk.nd;
    const int row0 = wk.first;
    const int sh = g.sh, tid = threadIdx.x;
    const int shv = VEC == 2 ? sh - 1 : sh;                      // log2 of the lanes per row part
    const int a = (tid & ((g.TP / VEC) - 1)) * VEC, q = tid >> shv, Q = NB >> shv;
    const bool wide = (g.TP / VEC) > 64;                         // only with VEC == 1
    const int n = nd.n, m = n + nd.b;
    const double *__restrict__ Fp = f.F + (nd.foff << sh) + a;
    const double *__restrict__ W0 = f.W + (nd.woff << sh) + a;         // child 0's plane; child 1's is m rows further
    const int64_t plane = (int64_t)m << sh;
    const bool live = a < g.ncol;
    const int nr = min(rb, m - row0);

    Vd<VEC> acc[RB];
    const double *rowp[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[r].v[c] = 0.0;
        rowp[r] = Fp + (((int64_t)(r < nr ? row0 + r : row0) * n) << sh);
    }
    // where this thread's (first) update row goes in the parent's plane: loaded now, needed after the fold
    const int cm0 = (q < nr && row0 + q >= n) ? f.cmap[nd.bdoff + (row0 + q - n)] : 0;
    // rows of L^-1 only need the columns j <= i: the block's last row bounds the loop
    const int last = row0 + nr - 1;
    const int jmax = last < n ? last + 1 : n;
    if (live) {
        for (int j = q; j < jmax; j += Q) {
            const int64_t jo = (int64_t)j << sh;
            const int64_t row = VMAP ? (int64_t)f.vmap[nd.k0 + j] : (int64_t)(nd.k0 + j);
            const Vd<VEC> wb = vload<VEC>(bhat + (row << sh) + a);
            Vd<VEC> w0, w1, fv[RB];
            if (!LEAF) {
                w0 = vload<VEC>(W0 + jo);
                w1 = vload<VEC>(W0 + plane + jo);
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) fv[r] = vload<VEC>(rowp[r] + jo);
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const double w = LEAF ? wb.v[c] : wb.v[c] - (w0.v[c] + w1.v[c]);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r].v[c] += fv[r].v[c] * w;
            }
        }
    }
    if (VEC == 1 && wide) front_fold_wide<NB, RB>(reinterpret_cast<Vd<1>(&)[RB]>(acc), red, tid);
    else front_fold<NB, RB, VEC>(acc, red, g.TP, a, tid);
    for (int r = q; r < nr && live; r += Q) {
        Vd<VEC> s;
        if (VEC == 1 && wide) s.v[0] = front_folded_wide<NB>(red, r, sh, a).v[0];
        else s = front_folded<NB, VEC>(red, r, g.TP, a);
        const int i = row0 + r;
        if (i < n) {
            vstore<VEC>(Y + (front_row(f, nd.k0 + i) << sh) + a, s);
        } else {   // update row: carry the children's contributions on, hand the sum to the parent's plane
            if (!LEAF) {
                const Vd<VEC> c0 = vload<VEC>(W0 + ((int64_t)i << sh)), c1 = vload<VEC>(W0 + plane + ((int64_t)i << sh));
#pragma unroll
                for (int c = 0; c < VEC; ++c) s.v[c] += c0.v[c] + c1.v[c];
            }
            const int cm = r == q ? cm0 : f.cmap[nd.bdoff + (i - n)];
            vstore<VEC>(f.W + ((nd.parent_w + cm) << sh) + a, s);
        }
    }
}

// backward sweep of one tree height.  Workgroup = (node, cb <= RB columns).  Same load discipline.
template <int NB, int RB, bool VMAP, int VEC>
__global__ __launch_bounds__(NB) void k_front_bwd(FrontArgs g, FrontDev f, const FrontWork *__restrict__ desc, int cb, const double *__restrict__ Y,
                                                  double *X) {
    __shared__ double red[RB * (NB / 64) * 64 * VEC];
    const FrontWork wk = desc[blockIdx.x];
    const FrontNode &nd = wk.nd;
    const int col0 = wk.first;
    const int sh = g.sh, tid = threadIdx.x;
    const int shv = VEC == 2 ? sh - 1 : sh;
    const int a = (tid & ((g.TP / VEC) - 1)) * VEC, q = tid >> shv, Q = NB >> shv;
    const bool wide = (g.TP / VEC) > 64;
    const int n = nd.n, m = n + nd.b;
    const double *__restrict__ Fp = f.F + (nd.foff << sh) + a;
    const int *__restrict__ bdv = f.bd_vertex + nd.bdoff;
    const bool live = a < g.ncol;
    const int nc = min(cb, n - col0);

    Vd<VEC> acc[RB];
    int64_t co[RB];      // column offsets (columns past the block: its first column, sums dropped)
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[r].v[c] = 0.0;
        co[r] = (int64_t)(r < nc ? col0 + r : col0) << sh;
    }
    if (live) {
        // rows of the separator: y_p.  Column i of L^-1 is zero above the diagonal: start at the block's first column
        for (int j = col0 + q; j < n; j += Q) {
            const int64_t row = VMAP ? (int64_t)f.vmap[nd.k0 + j] : (int64_t)(nd.k0 + j);
            const Vd<VEC> v = vload<VEC>(Y + (row << sh) + a);
            const double *__restrict__ Fj = Fp + (((int64_t)j * n) << sh);
            Vd<VEC> fv[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) fv[r] = vload<VEC>(Fj + co[r]);
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc[r].v[c] += fv[r].v[c] * v.v[c];
        }
        // boundary rows: -x of the ancestors (written by the launches of greater heights)
        for (int j = n + q; j < m; j += Q) {
            const Vd<VEC> v = vload<VEC>(X + ((int64_t)bdv[j - n] << sh) + a);
            const double *__restrict__ Fj = Fp + (((int64_t)j * n) << sh);
            Vd<VEC> fv[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) fv[r] = vload<VEC>(Fj + co[r]);
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc[r].v[c] -= fv[r].v[c] * v.v[c];
        }
    }
    if (VEC == 1 && wide) front_fold_wide<NB, RB>(reinterpret_cast<Vd<1>(&)[RB]>(acc), red, tid);
    else front_fold<NB, RB, VEC>(acc, red, g.TP, a, tid);
    for (int r = q; r < nc && live; r += Q) {
        Vd<VEC> s;
        if (VEC == 1 && wide) s.v[0] = front_folded_wide<NB>(red, r, sh, a).v[0];
        else s = front_folded<NB, VEC>(red, r, g.TP, a);
        vstore<VEC>(X + (front_row(f, nd.k0 + col0 + r) << sh) + a, s);
    }
}

void front_release(Ctx *c) {
    for (int i = 0; i < c->n_front_allocs; ++i) (void)hipFree(c->front_allocs[i]);
    c->n_front_allocs = 0;
    c->front = FrontDev{};
    c->use_front = 0;
    c->front_bytes = 0.0;
}

int front_setup(Ctx *c, const dots_front_desc *h) {
    const Dev &d = c->dcg;
    auto bad = [&](const char *what) {
        set_error(std::string("front_setup: ") + what);
        return (int)DOTS_ERR_ARGUMENT;
    };
    if (!h || h->n_nodes < 1 || h->n_levels < 1 || h->n_levels > 64) return bad("bad description");
    if (!h->node_n || !h->node_b || !h->node_foff || !h->node_ioff || !h->node_uoff || !h->node_child || !h->front_idx || !h->pull0 ||
        !h->pull1 || !h->level_ptr || !h->level_nodes || (!h->values && !h->grounded))
        return bad("null array");
    if (h->pitch != d.TP || h->n_modes != d.cg_ncol) return bad("pitch / mode count does not match the context");
    // ---- index sanity: a wrong index would fault on the device -------------------------------------
    const int nn = h->n_nodes;
    int64_t fo = 0, io = 0, uo = 0, eliminated = 0;
    double entries_read = 0.0;
    for (int p = 0; p < nn; ++p) {
        const int64_t n = h->node_n[p], b = h->node_b[p];
        if (n < 0 || b < 0 || n + b < 1) return bad("node size");
        if (h->node_foff[p] != fo || h->node_ioff[p] != io || h->node_uoff[p] != uo) return bad("node offsets are not the running sums");
        for (int k = 0; k < 2; ++k) {
            const int ch = h->node_child[2 * p + k];
            if (ch < -1 || ch >= p) return bad("child index (nodes must be numbered children first)");
        }
        const int c0 = h->node_child[2 * p], c1 = h->node_child[2 * p + 1];
        for (int64_t i = 0; i < n + b; ++i) {
            const int v = h->front_idx[io + i];
            if (v < 0 || v >= d.V) return bad("front vertex out of range");
            const int k0 = h->pull0[io + i], k1 = h->pull1[io + i];
            if (k0 < -1 || (k0 >= 0 && (c0 < 0 || k0 >= h->node_b[c0]))) return bad("pull0 out of range");
            if (k1 < -1 || (k1 >= 0 && (c1 < 0 || k1 >= h->node_b[c1]))) return bad("pull1 out of range");
        }
        fo += (n + b) * n;
        io += n + b;
        uo += b;
        eliminated += n;
        entries_read += 0.5 * n * (n + 1) + (double)b * n;
    }
    if (fo != h->n_entries || io != h->n_front_rows || uo != h->update_rows || eliminated != d.V) return bad("totals do not match");
    if (h->level_ptr[0] != 0 || h->level_ptr[h->n_levels] != nn) return bad("level_ptr");
    std::vector<int> level_of(nn, -1);
    for (int l = 0; l < h->n_levels; ++l) {
        if (h->level_ptr[l + 1] < h->level_ptr[l]) return bad("level_ptr not monotone");
        for (int k = h->level_ptr[l]; k < h->level_ptr[l + 1]; ++k) {
            const int p = h->level_nodes[k];
            if (p < 0 || p >= nn || level_of[p] != -1) return bad("level_nodes is not a permutation");
            level_of[p] = l;
        }
    }
    for (int p = 0; p < nn; ++p)
        for (int k = 0; k < 2; ++k) {
            const int ch = h->node_child[2 * p + k];
            if (ch >= 0 && level_of[ch] >= level_of[p]) return bad("a child is not on a lower level than its parent");
        }

    DOTS_HIP(hipStreamSynchronize(c->stream));
    front_release(c);
    // ---- node records, elimination order, update planes ------------------------------------------------
    std::vector<FrontNode> nodes((size_t)nn);
    std::vector<int> vmap((size_t)d.V), bd_vertex((size_t)std::max<int64_t>(h->update_rows, 1)), cmap((size_t)std::max<int64_t>(h->update_rows, 1), 0);
    std::vector<char> seen((size_t)d.V, 0);
    // W starts with a pad of zeros as long as the two planes of the largest front: nodes without children read their
    // (absent) children's updates there, so the sweeps need no branch on "has children"
    int max_front = 1;
    for (int p = 0; p < nn; ++p) max_front = std::max(max_front, h->node_n[p] + h->node_b[p]);
    int64_t wrows = 2 * (int64_t)max_front;
    {
        int k0 = 0;
        int64_t soff = 0;
        bool identity = true;
        for (int p = 0; p < nn; ++p) {
            FrontNode &nd = nodes[(size_t)p];
            nd.n = h->node_n[p];
            nd.b = h->node_b[p];
            nd.k0 = k0;
            nd.foff = h->node_foff[p];
            nd.bdoff = h->node_uoff[p];
            nd.parent_w = -1;
            nd.c0 = h->node_child[2 * p];
            nd.c1 = h->node_child[2 * p + 1];
            nd.ioff = h->node_ioff[p];
            nd.soff = soff;
            soff += (int64_t)nd.b * nd.b;
            nd.has_children = (h->node_child[2 * p] >= 0 || h->node_child[2 * p + 1] >= 0) ? 1 : 0;
            nd.woff = nd.has_children ? wrows : 0;
            if (nd.has_children) wrows += 2 * (int64_t)(nd.n + nd.b);
            const int64_t io = h->node_ioff[p];
            for (int i = 0; i < nd.n; ++i) {
                const int v = h->front_idx[io + i];
                if (seen[(size_t)v]) return bad("a vertex is eliminated twice");
                seen[(size_t)v] = 1;
                vmap[(size_t)(k0 + i)] = v;
                identity = identity && v == k0 + i;
            }
            for (int i = 0; i < nd.b; ++i) bd_vertex[(size_t)(nd.bdoff + i)] = h->front_idx[io + nd.n + i];
            k0 += nd.n;
        }
        // the plane a child writes and where its boundary rows sit in the parent's front (from the pull maps)
        for (int p = 0; p < nn; ++p) {
            const FrontNode &nd = nodes[(size_t)p];
            const int64_t io = h->node_ioff[p];
            const int m = nd.n + nd.b;
            for (int k = 0; k < 2; ++k) {
                const int ch = h->node_child[2 * p + k];
                if (ch < 0) continue;
                FrontNode &cn = nodes[(size_t)ch];
                if (cn.parent_w != -1) return bad("a node has two parents");
                cn.parent_w = nd.woff + (int64_t)k * m;
                const int32_t *pull = k == 0 ? h->pull0 : h->pull1;
                std::vector<char> got((size_t)cn.b, 0);
                for (int fpos = 0; fpos < m; ++fpos) {
                    const int r = pull[io + fpos];
                    if (r < 0) continue;
                    if (got[(size_t)r]) return bad("a child boundary row is pulled twice");
                    got[(size_t)r] = 1;
                    cmap[(size_t)(cn.bdoff + r)] = fpos;
                }
                for (int r = 0; r < cn.b; ++r)
                    if (!got[(size_t)r]) return bad("a child boundary row is not pulled by its parent");
            }
        }
        for (int p = 0; p < nn; ++p)
            if (nodes[(size_t)p].parent_w == -1 && nodes[(size_t)p].b != 0) return bad("a node without parent has boundary rows");
        if (identity) vmap.clear();
    }

    // ---- workgroup lists: (node, first row) per level for the forward sweep, (node, first column) backward.
    // Levels with many rows: 256-thread workgroups of up to 4 rows; the few large nodes near the root: one row
    // per 1024-thread workgroup, so that the long dot products are split 4x finer.
    std::vector<FrontWork> fwd, bwd;
    auto work = [&](int p, int first) {
        FrontWork w{};
        w.nd = nodes[(size_t)p];
        w.first = first;
        return w;
    };
    for (int l = 0; l < h->n_levels; ++l) {
        int64_t rows = 0, cols = 0;
        for (int k = h->level_ptr[l]; k < h->level_ptr[l + 1]; ++k) {
            const int p = h->level_nodes[k];
            rows += h->node_n[p] + h->node_b[p];
            cols += h->node_n[p];
        }
        // keep >= ~1000 workgroups where the level allows; at the bottom of the tree (tens of thousands of rows in
        // small nodes) up to 16 rows per workgroup: the fixed latency chain of a workgroup (descriptor, node
        // record, loads, fold, store) then covers 4x more bytes
        const int rbmax = c->front_rb_max;
        auto block = [rbmax](int64_t total) {
            int b = total >= 4096 ? 4 : (total >= 2048 ? 2 : 1);
            return std::min(b, rbmax);
        };
        const int rb = block(rows), cb = block(cols);
        c->front_fwd_rb[l] = rb;
        c->front_bwd_cb[l] = cb;
        c->front_fwd_nb[l] = (rows < 1024 && d.TP <= 256) ? 1024 : 256;
        c->front_bwd_nb[l] = (cols < 1024 && d.TP <= 256) ? 1024 : 256;
        c->front_fwd_ptr[l] = (int)fwd.size();
        c->front_bwd_ptr[l] = (int)bwd.size();
        for (int k = h->level_ptr[l]; k < h->level_ptr[l + 1]; ++k) {
            const int p = h->level_nodes[k];
            for (int r = 0; r < h->node_n[p] + h->node_b[p]; r += rb) fwd.push_back(work(p, r));
            for (int r = 0; r < h->node_n[p]; r += cb) bwd.push_back(work(p, r));
        }
    }
    c->front_fwd_ptr[h->n_levels] = (int)fwd.size();
    c->front_bwd_ptr[h->n_levels] = (int)bwd.size();

    FrontDev f{};
    f.n_nodes = nn;
    f.n_levels = h->n_levels;
    int rc;
#define FUP(field, src, n) if ((rc = front_upload(c, &f.field, src, (int64_t)(n)))) { front_release(c); return rc; }
    FUP(nodes, nodes.data(), nn);
    if (!vmap.empty()) FUP(vmap, vmap.data(), d.V);
    FUP(bd_vertex, bd_vertex.data(), bd_vertex.size()); FUP(cmap, cmap.data(), cmap.size());
    if (h->values) {
        FUP(F, h->values, h->n_entries << d.tp_shift);
    } else {   // numeric factorisation on the device (kernels_factor.hip)
        const double *t = nullptr;
        if ((rc = front_upload<double>(c, &t, nullptr, h->n_entries << d.tp_shift))) { front_release(c); return rc; }
        std::vector<int> grounded((size_t)d.TP, 0);
        for (int a = 0; a < h->n_modes; ++a) grounded[(size_t)a] = h->grounded[a] ? 1 : 0;
        if ((rc = front_factorize(c, h, f, nodes, const_cast<double *>(t), grounded.data()))) { front_release(c); return rc; }
        f.F = t;
    }
    FUP(fwd_desc, fwd.data(), std::max<size_t>(fwd.size(), 1)); FUP(bwd_desc, bwd.data(), std::max<size_t>(bwd.size(), 1));
    const double *w = nullptr;
    if ((rc = front_upload<double>(c, &w, nullptr, std::max<int64_t>(wrows, 1) << d.tp_shift))) { front_release(c); return rc; }
    f.W = const_cast<double *>(w);
#undef FUP
    c->front = f;
    c->use_front = 1;
    c->front_bytes = 2.0 * entries_read * d.cg_ncol * sizeof(double);
    return 0;
}

int front_solve(Ctx *c, const double *bhat, double *y, double *x) {
    const Dev &d = c->dcg;
    const FrontDev &f = c->front;
    if (f.n_nodes == 0) { set_error("front_solve: no factor installed"); return DOTS_ERR_STATE; }
    const FrontArgs g{d.tp_shift, d.TP, d.cg_ncol};
    const bool vm = f.vmap != nullptr;
    // two modes per lane (16-byte loads): pays where the sweeps are bandwidth-bound (measured: +6 % at torus100k, +13 % at
    // T = 127; -2 % on the latency-bound sphere10k, where it is left off)
    const bool v2 = c->front_vec2 && d.TP >= 4 && d.TP <= 128 && (c->front_vec2 > 1 || d.TP >= 64 || c->front_bytes > 1.0e9);
#define FRONT_FWD(NBV, RBV, LEAFV)                                                                                                 \
    do {                                                                                                                           \
        if (v2) {                                                                                                                  \
            if (vm) hipLaunchKernelGGL((k_front_fwd<NBV, RBV, true, LEAFV, 2>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, bhat, y);  \
            else hipLaunchKernelGGL((k_front_fwd<NBV, RBV, false, LEAFV, 2>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, bhat, y);    \
        } else {                                                                                                                   \
            if (vm) hipLaunchKernelGGL((k_front_fwd<NBV, RBV, true, LEAFV, 1>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, bhat, y);  \
            else hipLaunchKernelGGL((k_front_fwd<NBV, RBV, false, LEAFV, 1>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, bhat, y);    \
        }                                                                                                                          \
    } while (0)
#define FRONT_BWD(NBV, RBV)                                                                                                        \
    do {                                                                                                                           \
        if (v2) {                                                                                                                  \
            if (vm) hipLaunchKernelGGL((k_front_bwd<NBV, RBV, true, 2>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, y, x);  \
            else hipLaunchKernelGGL((k_front_bwd<NBV, RBV, false, 2>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, y, x);    \
        } else {                                                                                                                   \
            if (vm) hipLaunchKernelGGL((k_front_bwd<NBV, RBV, true, 1>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, y, x);  \
            else hipLaunchKernelGGL((k_front_bwd<NBV, RBV, false, 1>), dim3(n), dim3(NBV), 0, c->stream, g, f, ptr, blk, y, x);    \
        }                                                                                                                          \
    } while (0)
    for (int l = 0; l < f.n_levels; ++l) {
        const int n = c->front_fwd_ptr[l + 1] - c->front_fwd_ptr[l];
        if (n <= 0) continue;
        const FrontWork *ptr = f.fwd_desc + c->front_fwd_ptr[l];
        const int blk = c->front_fwd_rb[l];
        const bool big = c->front_fwd_nb[l] == 1024;
        if (l == 0) {            // height 0: leaves only
            if (big) { if (blk == 1) FRONT_FWD(1024, 1, true); else if (blk == 2) FRONT_FWD(1024, 2, true); else FRONT_FWD(1024, 4, true); }
            else { if (blk == 1) FRONT_FWD(256, 1, true); else if (blk == 2) FRONT_FWD(256, 2, true); else FRONT_FWD(256, 4, true); }
        } else {
            if (big) { if (blk == 1) FRONT_FWD(1024, 1, false); else if (blk == 2) FRONT_FWD(1024, 2, false); else FRONT_FWD(1024, 4, false); }
            else { if (blk == 1) FRONT_FWD(256, 1, false); else if (blk == 2) FRONT_FWD(256, 2, false); else FRONT_FWD(256, 4, false); }
        }
    }
    for (int l = f.n_levels - 1; l >= 0; --l) {
        const int n = c->front_bwd_ptr[l + 1] - c->front_bwd_ptr[l];
        if (n <= 0) continue;
        const FrontWork *ptr = f.bwd_desc + c->front_bwd_ptr[l];
        const int blk = c->front_bwd_cb[l];
        if (c->front_bwd_nb[l] == 1024) { if (blk == 1) FRONT_BWD(1024, 1); else if (blk == 2) FRONT_BWD(1024, 2); else FRONT_BWD(1024, 4); }
        else { if (blk == 1) FRONT_BWD(256, 1); else if (blk == 2) FRONT_BWD(256, 2); else FRONT_BWD(256, 4); }
    }
#undef FRONT_FWD
#undef FRONT_BWD
    DOTS_HIP(hipGetLastError());
    return 0;
}

}  // namespace dots
