// Host-side setup of the direct solver: geometric nested dissection of the mesh graph and the symbolic structure of
// the multifrontal factor (no device code; the same algorithms as dots_socp_amd/frontal.py, which remains the
// reference implementation for the tests).  At 10^5 vertices the Python recursion costs ~0.4 s, this ~0.03 s.
#include "dots_dev.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <vector>

struct dots_tree {
    std::vector<int64_t> order, sep_ptr;
    std::vector<int32_t> child, parent, height;
};
struct dots_mesh_ops {      // dots_assemble: the outputs of the reference's operator assembly as flat arrays
    std::vector<double> area, hat, mass, val;
    std::vector<int32_t> cptr, cidx, rowptr, col;
};
struct dots_symbolic {
    std::vector<int32_t> node_b, front_idx, pull0, pull1;
    int64_t update_rows = 0;
};

namespace {

struct Dissector {
    int pca_min = 0;                // subsets larger than this are cut across their principal axis, smaller ones across the longest bounding-box side
    int V, leaf;
    const int32_t *indptr, *indices;
    const double *xyz;
    std::vector<int> work;          // vertex ids; a subset is a contiguous range of it
    std::vector<signed char> mark;
    std::vector<std::pair<double, int>> key;
    std::vector<int> tmp;
    dots_tree *out;

    int emit(int lo, int hi, int c0, int c1) {          // the vertices work[lo:hi) are eliminated at a new node
        for (int i = lo; i < hi; ++i) out->order.push_back(work[i]);
        out->sep_ptr.push_back((int64_t)out->order.size());
        out->child.push_back(c0);
        out->child.push_back(c1);
        return (int)out->sep_ptr.size() - 2;
    }

    // direction of the cut: longest side of the bounding box; for large subsets the principal axis (power iteration)
    void direction(int lo, int hi, double dir[3], double box[3], double dir2[3]) {
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, mean[3] = {0, 0, 0};
        for (int i = lo; i < hi; ++i)
            for (int c = 0; c < 3; ++c) {
                const double x = xyz[3 * (size_t)work[i] + c];
                mn[c] = std::min(mn[c], x);
                mx[c] = std::max(mx[c], x);
                mean[c] += x;
            }
        int ax = 0;
        for (int c = 1; c < 3; ++c)
            if (mx[c] - mn[c] > mx[ax] - mn[ax]) ax = c;
        dir[0] = dir[1] = dir[2] = 0.0;
        dir[ax] = 1.0;
        box[0] = box[1] = box[2] = 0.0;
        box[ax] = 1.0;
        dir2[0] = dir2[1] = dir2[2] = 0.0;
        dir2[ax] = 1.0;
        if (hi - lo <= pca_min) return;
        for (int c = 0; c < 3; ++c) mean[c] /= (hi - lo);
        double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int i = lo; i < hi; ++i) {
            double d[3];
            for (int c = 0; c < 3; ++c) d[c] = xyz[3 * (size_t)work[i] + c] - mean[c];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) C[a][b] += d[a] * d[b];
        }
        for (int it = 0; it < 40; ++it) {
            double w[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) w[a] += C[a][b] * dir[b];
            const double nrm = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            if (!(nrm > 0.0)) break;
            for (int a = 0; a < 3; ++a) dir[a] = w[a] / nrm;
        }
        // second principal axis: power iteration in the complement of the first
        double v[3] = {box[1] + 0.3, box[2] + 0.2, box[0] + 0.1};
        for (int it = 0; it < 40; ++it) {
            double dot = v[0] * dir[0] + v[1] * dir[1] + v[2] * dir[2];
            for (int a = 0; a < 3; ++a) v[a] -= dot * dir[a];
            double w[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) w[a] += C[a][b] * v[b];
            dot = w[0] * dir[0] + w[1] * dir[1] + w[2] * dir[2];
            for (int a = 0; a < 3; ++a) w[a] -= dot * dir[a];
            const double nrm = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            if (!(nrm > 0.0)) return;
            for (int a = 0; a < 3; ++a) v[a] = w[a] / nrm;
        }
        for (int a = 0; a < 3; ++a) dir2[a] = v[a];
    }

    int build(int lo, int hi) {
        const int n = hi - lo;
        if (n <= leaf) return emit(lo, hi, -1, -1);
        // candidate directions (the two principal axes, the longest bounding-box side): keep the smallest separator
        double dirs[3][3];
        direction(lo, hi, dirs[0], dirs[1], dirs[2]);
        const int half = n / 2;
        tmp.assign(work.begin() + lo, work.begin() + hi);
        auto split = [&](const double *dir) {
            key.resize((size_t)n);
            for (int i = 0; i < n; ++i) {
                const double *p = xyz + 3 * (size_t)tmp[(size_t)i];
                key[(size_t)i] = {p[0] * dir[0] + p[1] * dir[1] + p[2] * dir[2], i};     // ties: by position in the subset
            }
            std::nth_element(key.begin(), key.begin() + half, key.end());
            for (int i = 0; i < n; ++i) {
                work[lo + i] = tmp[(size_t)key[(size_t)i].second];
                mark[(size_t)work[lo + i]] = i < half ? 1 : 2;
            }
            int cA = 0, cB = 0;
            for (int i = 0; i < half; ++i)
                for (int k = indptr[work[lo + i]]; k < indptr[work[lo + i] + 1]; ++k)
                    if (mark[(size_t)indices[k]] == 2) { ++cA; break; }
            for (int i = half; i < n; ++i)
                for (int k = indptr[work[lo + i]]; k < indptr[work[lo + i] + 1]; ++k)
                    if (mark[(size_t)indices[k]] == 1) { ++cB; break; }
            return std::min(cA, cB);
        };
        int best = 0, best_size = split(dirs[0]), last = 0;
        for (int k = 1; k < 3; ++k) {
            bool dup = false;
            for (int q = 0; q < k; ++q) dup = dup || (dirs[k][0] == dirs[q][0] && dirs[k][1] == dirs[q][1] && dirs[k][2] == dirs[q][2]);
            if (dup) continue;
            const int sz = split(dirs[k]);
            last = k;
            if (sz < best_size) { best_size = sz; best = k; }
        }
        if (best != last) split(dirs[best]);       // leave work / mark in the state of the best cut
        auto touches = [&](int v, signed char other) {
            for (int k = indptr[v]; k < indptr[v + 1]; ++k)
                if (mark[(size_t)indices[k]] == other) return true;
            return false;
        };
        int nA = 0, nB = 0;
        for (int i = 0; i < half; ++i) nA += touches(work[lo + i], 2);
        for (int i = half; i < n; ++i) nB += touches(work[lo + i], 1);
        // separator = the smaller one-sided boundary; arrange the range as [A' | B' | separator]
        const bool fromA = nA <= nB;
        std::vector<int> a, b, s;
        for (int i = 0; i < n; ++i) {
            const int v = work[lo + i];
            const bool inA = i < half;
            const bool sep = inA == fromA && touches(v, inA ? 2 : 1);
            (sep ? s : (inA ? a : b)).push_back(v);
        }
        for (int i = 0; i < n; ++i) mark[(size_t)work[lo + i]] = 0;
        if (a.empty() || b.empty()) return emit(lo, hi, -1, -1);      // degenerate cut: eliminate the subset densely
        std::copy(a.begin(), a.end(), work.begin() + lo);
        std::copy(b.begin(), b.end(), work.begin() + lo + (int)a.size());
        std::copy(s.begin(), s.end(), work.begin() + lo + (int)a.size() + (int)b.size());
        const int na = (int)a.size(), nb = (int)b.size();
        a.clear(); a.shrink_to_fit(); b.clear(); b.shrink_to_fit(); s.clear(); s.shrink_to_fit();
        const int c0 = build(lo, lo + na);
        const int c1 = build(lo + na, lo + na + nb);
        return emit(lo + na + nb, hi, c0, c1);
    }
};

}  // namespace

namespace {
// recursive coordinate bisection of the points ids[lo:hi): longest side of the bounding box, cut at a multiple of `unit`
void patch_bisect(const double *xyz, int32_t *ids, int lo, int hi, int unit) {
    const int m = hi - lo;
    if (m <= unit) return;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int i = lo; i < hi; ++i)
        for (int c = 0; c < 3; ++c) {
            const double x = xyz[3 * (size_t)ids[i] + c];
            mn[c] = std::min(mn[c], x);
            mx[c] = std::max(mx[c], x);
        }
    int ax = 0;
    for (int c = 1; c < 3; ++c)
        if (mx[c] - mn[c] > mx[ax] - mn[ax]) ax = c;
    int half = std::max(unit, (m / 2 + unit - 1) / unit * unit);
    if (half >= m) half = m - 1;
    std::nth_element(ids + lo, ids + lo + half, ids + hi, [&](int32_t a, int32_t b) {
        const double xa = xyz[3 * (size_t)a + ax], xb = xyz[3 * (size_t)b + ax];
        return xa != xb ? xa < xb : a < b;
    });
    patch_bisect(xyz, ids, lo, lo + half, unit);
    patch_bisect(xyz, ids, lo + half, hi, unit);
}
}  // namespace

extern "C" {

// Operator assembly (reference: utils/surface_pre_computations_socp.py:11-132, socp/solver_socp.py:102-113): triangle areas
// (:25), hat-function gradients (:31-37), vertex masses (:112), the vertex -> corner lists in the reference's corner order
// i = k F + f (:114-118) and K = G^T diag(area) G = minus the cotangent matrix (:68-84) as a sorted CSR.  The same formulas, in
// the same order of operations, as dots_socp_amd/geometry.py (hat_gradients, corner_lists, stiffness_matrix: the reference
// implementations the tests compare with); one pass over the triangles and one over the corner lists instead of numpy
// temporaries and a sparse product: 0.01 s instead of 0.2 s at 10^5 vertices.
int dots_assemble(int32_t V, int32_t F, const double *xyz, const int32_t *tri, dots_mesh_ops **out) {
    if (V < 1 || F < 1 || !xyz || !tri || !out) { dots::set_error("assemble: bad argument"); return DOTS_ERR_ARGUMENT; }
    for (int64_t i = 0; i < (int64_t)3 * F; ++i)
        if (tri[i] < 0 || tri[i] >= V) { dots::set_error("assemble: triangle index out of range"); return DOTS_ERR_ARGUMENT; }
    dots_mesh_ops *m = new dots_mesh_ops();
    m->area.resize((size_t)F);
    m->hat.resize((size_t)F * 9);
    for (int f = 0; f < F; ++f) {
        double p[3][3];
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 3; ++c) p[k][c] = xyz[3 * (size_t)tri[3 * f + k] + c];
        double u[3], w2[3];
        for (int c = 0; c < 3; ++c) { u[c] = p[1][c] - p[0][c]; w2[c] = p[2][c] - p[1][c]; }
        const double cx = u[1] * w2[2] - u[2] * w2[1], cy = u[2] * w2[0] - u[0] * w2[2], cz = u[0] * w2[1] - u[1] * w2[0];
        m->area[(size_t)f] = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
        for (int k = 0; k < 3; ++k) {       // altitude from the opposite edge to corner k, over its squared length
            const double *a = p[(k + 1) % 3], *b = p[(k + 2) % 3];
            double e[3], w[3], alt[3];
            for (int c = 0; c < 3; ++c) { e[c] = b[c] - a[c]; w[c] = p[k][c] - a[c]; }
            const double we = w[0] * e[0] + w[1] * e[1] + w[2] * e[2], ee = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
            const double q = we / ee;
            for (int c = 0; c < 3; ++c) alt[c] = w[c] - e[c] * q;
            const double aa = alt[0] * alt[0] + alt[1] * alt[1] + alt[2] * alt[2];
            for (int c = 0; c < 3; ++c) m->hat[((size_t)f * 3 + k) * 3 + c] = alt[c] / aa;
        }
    }
    // corner lists, ordered by (k, f) within a vertex; masses in the same order of summation
    m->cptr.assign((size_t)V + 1, 0);
    for (int64_t i = 0; i < (int64_t)3 * F; ++i) m->cptr[(size_t)tri[i] + 1] += 1;
    for (int v = 0; v < V; ++v) m->cptr[(size_t)v + 1] += m->cptr[(size_t)v];
    m->cidx.resize((size_t)3 * F);
    m->mass.assign((size_t)V, 0.0);
    {
        std::vector<int32_t> fill(m->cptr.begin(), m->cptr.end() - 1);
        for (int k = 0; k < 3; ++k)
            for (int f = 0; f < F; ++f) {
                const int v = tri[3 * f + k];
                m->cidx[(size_t)fill[(size_t)v]++] = f * 3 + k;
                m->mass[(size_t)v] += m->area[(size_t)f];
            }
    }
    for (int v = 0; v < V; ++v) m->mass[(size_t)v] /= 3.0;
    // K, row by row: the contributions of a vertex's corners in list order, columns sorted
    m->rowptr.assign((size_t)V + 1, 0);
    std::vector<std::pair<int32_t, double>> row;
    for (int v = 0; v < V; ++v) {
        row.clear();
        for (int j = m->cptr[(size_t)v]; j < m->cptr[(size_t)v + 1]; ++j) {
            const int fk = m->cidx[(size_t)j], f = fk / 3;
            const double *gk = &m->hat[(size_t)fk * 3];
            for (int i = 0; i < 3; ++i) {
                const double *gi = &m->hat[((size_t)f * 3 + i) * 3];
                const double val = m->area[(size_t)f] * (gk[0] * gi[0] + gk[1] * gi[1] + gk[2] * gi[2]);
                const int32_t u = tri[3 * f + i];
                size_t at = 0;
                while (at < row.size() && row[at].first != u) ++at;
                if (at == row.size()) row.emplace_back(u, val);
                else row[at].second += val;
            }
        }
        std::sort(row.begin(), row.end(), [](const std::pair<int32_t, double> &x, const std::pair<int32_t, double> &y) { return x.first < y.first; });
        for (const auto &e : row) { m->col.push_back(e.first); m->val.push_back(e.second); }
        m->rowptr[(size_t)v + 1] = (int32_t)m->col.size();
    }
    *out = m;
    return 0;
}
int64_t dots_assemble_nnz(const dots_mesh_ops *m) { return m ? (int64_t)m->col.size() : -1; }
int dots_assemble_copy(const dots_mesh_ops *m, double *area, double *hat, double *mass, int32_t *cptr, int32_t *cidx, int32_t *rowptr, int32_t *col, double *val) {
    if (!m || !area || !hat || !mass || !cptr || !cidx || !rowptr || !col || !val) { dots::set_error("assemble_copy: null argument"); return DOTS_ERR_ARGUMENT; }
    std::copy(m->area.begin(), m->area.end(), area);
    std::copy(m->hat.begin(), m->hat.end(), hat);
    std::copy(m->mass.begin(), m->mass.end(), mass);
    std::copy(m->cptr.begin(), m->cptr.end(), cptr);
    std::copy(m->cidx.begin(), m->cidx.end(), cidx);
    std::copy(m->rowptr.begin(), m->rowptr.end(), rowptr);
    std::copy(m->col.begin(), m->col.end(), col);
    std::copy(m->val.begin(), m->val.end(), val);
    return 0;
}
void dots_assemble_free(dots_mesh_ops *m) { delete m; }

int dots_patch_order(int32_t n_vertices, const double *xyz, int32_t unit, int32_t *order) {
    if (n_vertices < 1 || !xyz || unit < 1 || !order) { dots::set_error("patch_order: bad argument"); return DOTS_ERR_ARGUMENT; }
    std::iota(order, order + n_vertices, 0);
    patch_bisect(xyz, order, 0, n_vertices, unit);
    return 0;
}

int dots_tree_build(int32_t n_vertices, const int32_t *indptr, const int32_t *indices, const double *xyz, int32_t leaf, dots_tree **out) {
    if (n_vertices < 1 || !indptr || !indices || !xyz || leaf < 1 || !out) { dots::set_error("tree_build: bad argument"); return DOTS_ERR_ARGUMENT; }
    for (int v = 0; v < n_vertices; ++v)
        if (indptr[v + 1] < indptr[v]) { dots::set_error("tree_build: indptr not monotone"); return DOTS_ERR_ARGUMENT; }
    for (int k = indptr[0]; k < indptr[n_vertices]; ++k)
        if (indices[k] < 0 || indices[k] >= n_vertices) { dots::set_error("tree_build: index out of range"); return DOTS_ERR_ARGUMENT; }
    dots_tree *t = new dots_tree();
    t->sep_ptr.push_back(0);
    Dissector d{};
    d.V = n_vertices; d.leaf = leaf; d.indptr = indptr; d.indices = indices; d.xyz = xyz; d.out = t;
    if (!dots::env_int("DOTS_ND_PCA_MIN", 0, 1 << 30, &d.pca_min)) { delete t; return DOTS_ERR_ARGUMENT; }
    d.work.resize((size_t)n_vertices);
    std::iota(d.work.begin(), d.work.end(), 0);
    d.mark.assign((size_t)n_vertices, 0);
    d.build(0, n_vertices);
    const int nn = (int)t->sep_ptr.size() - 1;
    t->parent.assign((size_t)nn, -1);
    t->height.assign((size_t)nn, 0);
    for (int p = 0; p < nn; ++p)
        for (int k = 0; k < 2; ++k) {
            const int c = t->child[2 * (size_t)p + k];
            if (c >= 0) {
                t->parent[(size_t)c] = p;
                t->height[(size_t)p] = std::max(t->height[(size_t)p], t->height[(size_t)c] + 1);
            }
        }
    *out = t;
    return 0;
}

int64_t dots_tree_nodes(const dots_tree *t) { return t ? (int64_t)t->sep_ptr.size() - 1 : -1; }

int dots_tree_copy(const dots_tree *t, int64_t *order, int64_t *sep_ptr, int32_t *child, int32_t *parent, int32_t *height) {
    if (!t || !order || !sep_ptr || !child || !parent || !height) { dots::set_error("tree_copy: null argument"); return DOTS_ERR_ARGUMENT; }
    std::copy(t->order.begin(), t->order.end(), order);
    std::copy(t->sep_ptr.begin(), t->sep_ptr.end(), sep_ptr);
    std::copy(t->child.begin(), t->child.end(), child);
    std::copy(t->parent.begin(), t->parent.end(), parent);
    std::copy(t->height.begin(), t->height.end(), height);
    return 0;
}

void dots_tree_free(dots_tree *t) { delete t; }

// Boundary sets, front index lists and pull maps of the tree (order, sep_ptr, child) on the graph (indptr, indices).
int dots_symbolic_build(int32_t n_vertices, const int32_t *indptr, const int32_t *indices, int64_t n_nodes, const int64_t *order, const int64_t *sep_ptr,
                        const int32_t *child, dots_symbolic **out) {
    if (n_vertices < 1 || !indptr || !indices || n_nodes < 1 || !order || !sep_ptr || !child || !out) { dots::set_error("symbolic_build: bad argument"); return DOTS_ERR_ARGUMENT; }
    if (sep_ptr[0] != 0 || sep_ptr[n_nodes] != n_vertices) { dots::set_error("symbolic_build: sep_ptr does not cover the vertices"); return DOTS_ERR_ARGUMENT; }
    std::vector<int64_t> pos((size_t)n_vertices, -1);
    for (int64_t k = 0; k < n_vertices; ++k) {
        if (order[k] < 0 || order[k] >= n_vertices || pos[(size_t)order[k]] != -1) { dots::set_error("symbolic_build: order is not a permutation"); return DOTS_ERR_ARGUMENT; }
        pos[(size_t)order[k]] = k;
    }
    for (int64_t p = 0; p < n_nodes; ++p)
        for (int k = 0; k < 2; ++k)
            if (child[2 * p + k] < -1 || child[2 * p + k] >= p) { dots::set_error("symbolic_build: child index"); return DOTS_ERR_ARGUMENT; }
    dots_symbolic *s = new dots_symbolic();
    std::vector<std::vector<int>> bd((size_t)n_nodes);
    std::vector<int64_t> stamp((size_t)n_vertices, -1);
    s->node_b.resize((size_t)n_nodes);
    for (int64_t p = 0; p < n_nodes; ++p) {
        const int64_t last = sep_ptr[p + 1];
        std::vector<int> &b = bd[(size_t)p];
        auto take = [&](int v) {
            if (pos[(size_t)v] >= last && stamp[(size_t)v] != p) {
                stamp[(size_t)v] = p;
                b.push_back(v);
            }
        };
        for (int64_t k = sep_ptr[p]; k < last; ++k) {
            const int v = (int)order[k];
            for (int e = indptr[v]; e < indptr[v + 1]; ++e) take(indices[e]);
        }
        for (int k = 0; k < 2; ++k) {
            const int c = child[2 * p + k];
            if (c >= 0)
                for (int v : bd[(size_t)c]) take(v);
        }
        std::sort(b.begin(), b.end(), [&](int x, int y) { return pos[(size_t)x] < pos[(size_t)y]; });
        s->node_b[(size_t)p] = (int32_t)b.size();
        s->update_rows += (int64_t)b.size();
    }
    // front index lists (separator first) and the position of every front row in the children's boundaries
    std::vector<int> frontpos((size_t)n_vertices, -1);
    for (int64_t p = 0; p < n_nodes; ++p) {
        const int64_t io = (int64_t)s->front_idx.size();
        for (int64_t k = sep_ptr[p]; k < sep_ptr[p + 1]; ++k) s->front_idx.push_back((int32_t)order[k]);
        for (int v : bd[(size_t)p]) s->front_idx.push_back(v);
        const int64_t m = (int64_t)s->front_idx.size() - io;
        s->pull0.resize((size_t)(io + m), -1);
        s->pull1.resize((size_t)(io + m), -1);
        for (int64_t i = 0; i < m; ++i) frontpos[(size_t)s->front_idx[(size_t)(io + i)]] = (int)i;
        for (int k = 0; k < 2; ++k) {
            const int c = child[2 * p + k];
            if (c < 0) continue;
            std::vector<int32_t> &pull = k == 0 ? s->pull0 : s->pull1;
            const std::vector<int> &cb = bd[(size_t)c];
            for (size_t r = 0; r < cb.size(); ++r) {
                const int fp = frontpos[(size_t)cb[r]];
                if (fp < 0) { delete s; dots::set_error("symbolic_build: a child's boundary vertex is missing from its parent's front"); return DOTS_ERR_STATE; }
                pull[(size_t)(io + fp)] = (int32_t)r;
            }
        }
        for (int64_t i = 0; i < m; ++i) frontpos[(size_t)s->front_idx[(size_t)(io + i)]] = -1;
    }
    *out = s;
    return 0;
}

int64_t dots_symbolic_front_rows(const dots_symbolic *s) { return s ? (int64_t)s->front_idx.size() : -1; }

int dots_symbolic_copy(const dots_symbolic *s, int32_t *node_b, int32_t *front_idx, int32_t *pull0, int32_t *pull1) {
    if (!s || !node_b || !front_idx || !pull0 || !pull1) { dots::set_error("symbolic_copy: null argument"); return DOTS_ERR_ARGUMENT; }
    std::copy(s->node_b.begin(), s->node_b.end(), node_b);
    std::copy(s->front_idx.begin(), s->front_idx.end(), front_idx);
    std::copy(s->pull0.begin(), s->pull0.end(), pull0);
    std::copy(s->pull1.begin(), s->pull1.end(), pull1);
    return 0;
}

void dots_symbolic_free(dots_symbolic *s) { delete s; }

}  // extern "C"
