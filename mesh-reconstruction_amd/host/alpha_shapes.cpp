// alpha_shapes.cpp -- alphaShapeFaces (recon.hpp:33-34, alpha_shapes.cpp:36-104 of the reference) without CGAL.
//
// The reference hands the bundle's points to CGAL (not in this image, not vendored by the reference: a system package):
//     Alpha_shape_3 as(points);                    // Delaunay triangulation, REGULARIZED mode (the constructor's default)
//     opt = as.find_optimal_alpha(1); as.set_alpha(*opt); *alpha = *opt;
//     faces = as.get_alpha_shape_facets(REGULAR)    // each oriented so that its normal points out of the solid
// What is restated here, from CGAL's published definitions (Alpha_shape_3.h, Delaunay_triangulation_3.h; CGAL 4.x / 5.x agree on them):
//   * Delaunay triangulation of the distinct points with EXACT predicates (the reference's kernel is
//     Exact_predicates_inexact_constructions_kernel): incremental Bowyer-Watson insertion with an infinite vertex; orient3d / insphere
//     are evaluated in long double behind a forward error bound and, when that cannot decide, exactly in 384-bit integers (the
//     coordinates are floats: scaled by one power of two they are integers).
//   * alpha of a cell = its squared circumradius, computed in double with CGAL's squared_radiusC3 formula (inexact constructions).
//   * REGULARIZED classification: a cell is interior iff it is finite and alpha_cell <= alpha; a facet is REGULAR iff exactly one of
//     its two cells is interior.
//   * find_alpha_solid(): max over vertices of the smallest alpha among the vertex's finite cells; the alpha spectrum: the distinct
//     positive cell alphas, ascending; number_of_solid_components(alpha): connected components (through facets) of interior cells;
//     find_optimal_alpha(n): lower bound of alpha_solid in the spectrum, then a binary search for the first value with <= n
//     components, and -- as CGAL does -- the spectrum entry AFTER that one when there is one.
// Not reproduced: CGAL's symbolic perturbation for five or more cospherical points (any valid Delaunay triangulation is produced;
// for float data from a bundle adjustment the case has measure zero), the order of the faces in the output and which vertex of a
// face comes first (alpha_shapes.cpp:87-95 depends on CGAL's memory layout there; the orientation does not).
// Tests: tests/test_meshing_cpu.py (scipy's Qhull Delaunay + a numpy restatement of the same definitions as the independent check).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "recon.hpp"

namespace {

// ---------------------------------------------------------------- exact arithmetic: 384-bit two's complement integers
struct Big {
    static constexpr int N = 6;
    uint64_t w[N];
    Big() { std::memset(w, 0, sizeof w); }
    explicit Big(__int128 v)
    {
        w[0] = (uint64_t)v;
        w[1] = (uint64_t)(v >> 64);
        const uint64_t ext = v < 0 ? ~0ull : 0ull;
        for (int i = 2; i < N; i++) w[i] = ext;
    }
    bool neg() const { return (w[N - 1] >> 63) != 0; }
    bool zero() const
    {
        for (int i = 0; i < N; i++)
            if (w[i]) return false;
        return true;
    }
    int sign() const { return zero() ? 0 : (neg() ? -1 : 1); }
    Big operator+(const Big &o) const
    {
        Big r;
        unsigned __int128 c = 0;
        for (int i = 0; i < N; i++) {
            c += (unsigned __int128)w[i] + o.w[i];
            r.w[i] = (uint64_t)c;
            c >>= 64;
        }
        return r;
    }
    Big operator-() const
    {
        Big r;
        unsigned __int128 c = 1;
        for (int i = 0; i < N; i++) {
            c += (unsigned __int128)(~w[i]);
            r.w[i] = (uint64_t)c;
            c >>= 64;
        }
        return r;
    }
    Big operator-(const Big &o) const { return *this + (-o); }
    Big operator*(const Big &o) const  // modulo 2^384: exact whenever the true product fits (it does: see the bit budgets at the callers)
    {
        Big r;
        for (int i = 0; i < N; i++) {
            unsigned __int128 c = 0;
            for (int j = 0; i + j < N; j++) {
                c += (unsigned __int128)w[i] * o.w[j] + r.w[i + j];
                r.w[i + j] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
};

struct P3 {
    double x, y, z;       // the point as CGAL sees it (float coordinates, held in double)
    int64_t ix, iy, iz;   // the same coordinates times one common power of two: integers of at most 62 bits
};

Big det3(const Big &a, const Big &b, const Big &c, const Big &d, const Big &e, const Big &f, const Big &g, const Big &h, const Big &i)
{
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}

// sign of det [b - a; c - a; d - a] = ((b - a) x (c - a)) . (d - a)
int orient3d(const P3 &a, const P3 &b, const P3 &c, const P3 &d)
{
    {
        const long double bx = (long double)b.ix - a.ix, by = (long double)b.iy - a.iy, bz = (long double)b.iz - a.iz;  // exact: <= 63 bits
        const long double cx = (long double)c.ix - a.ix, cy = (long double)c.iy - a.iy, cz = (long double)c.iz - a.iz;
        const long double dx = (long double)d.ix - a.ix, dy = (long double)d.iy - a.iy, dz = (long double)d.iz - a.iz;
        const long double m1 = cy * dz, m2 = cz * dy, m3 = cx * dz, m4 = cz * dx, m5 = cx * dy, m6 = cy * dx;
        const long double det = bx * (m1 - m2) - by * (m3 - m4) + bz * (m5 - m6);
        const long double perm = fabsl(bx) * (fabsl(m1) + fabsl(m2)) + fabsl(by) * (fabsl(m3) + fabsl(m4)) + fabsl(bz) * (fabsl(m5) + fabsl(m6));
        const long double bound = 16.0L * 5.42101086242752217e-20L * perm;  // 16 * 2^-64 * permanent: a generous forward bound for 11 roundings
        if (det > bound) return 1;
        if (det < -bound) return -1;
        if (perm == 0.0L) return 0;
    }
    auto D = [](int64_t p, int64_t q) { return Big((__int128)p - (__int128)q); };
    // entries < 2^64, products of three < 2^192, six of them < 2^195: fits 384 bits
    return det3(D(b.ix, a.ix), D(b.iy, a.iy), D(b.iz, a.iz), D(c.ix, a.ix), D(c.iy, a.iy), D(c.iz, a.iz), D(d.ix, a.ix), D(d.iy, a.iy), D(d.iz, a.iz)).sign();
}

// > 0 iff e lies strictly inside the sphere through a, b, c, d, for orient3d(a, b, c, d) > 0
int insphere(const P3 &a, const P3 &b, const P3 &c, const P3 &d, const P3 &e)
{
    const P3 *p[4] = {&a, &b, &c, &d};
    {
        long double x[4], y[4], z[4], l[4];
        for (int i = 0; i < 4; i++) {
            x[i] = (long double)p[i]->ix - e.ix;
            y[i] = (long double)p[i]->iy - e.iy;
            z[i] = (long double)p[i]->iz - e.iz;
            l[i] = x[i] * x[i] + y[i] * y[i] + z[i] * z[i];
        }
        // det | x y z l | expanded along the l column; m(i,j,k) = det of rows i, j, k of (x y z)
        auto m3 = [&](int i, int j, int k, long double &perm) {
            const long double t1 = y[j] * z[k], t2 = z[j] * y[k], t3 = x[j] * z[k], t4 = z[j] * x[k], t5 = x[j] * y[k], t6 = y[j] * x[k];
            perm = fabsl(x[i]) * (fabsl(t1) + fabsl(t2)) + fabsl(y[i]) * (fabsl(t3) + fabsl(t4)) + fabsl(z[i]) * (fabsl(t5) + fabsl(t6));
            return x[i] * (t1 - t2) - y[i] * (t3 - t4) + z[i] * (t5 - t6);
        };
        long double p0, p1, p2, p3;
        const long double d0 = m3(1, 2, 3, p0), d1 = m3(0, 2, 3, p1), d2 = m3(0, 1, 3, p2), d3 = m3(0, 1, 2, p3);
        // sign convention: for a positively oriented (a, b, c, d) the 4 x 4 determinant | x y z l | is NEGATIVE when e is inside
        const long double det = -(-l[0] * d0 + l[1] * d1 - l[2] * d2 + l[3] * d3);
        const long double perm = l[0] * p0 + l[1] * p1 + l[2] * p2 + l[3] * p3;
        const long double bound = 64.0L * 5.42101086242752217e-20L * perm;
        if (det > bound) return 1;
        if (det < -bound) return -1;
        if (perm == 0.0L) return 0;
    }
    Big x[4], y[4], z[4], l[4];
    for (int i = 0; i < 4; i++) {
        x[i] = Big((__int128)p[i]->ix - (__int128)e.ix);
        y[i] = Big((__int128)p[i]->iy - (__int128)e.iy);
        z[i] = Big((__int128)p[i]->iz - (__int128)e.iz);
        l[i] = x[i] * x[i] + y[i] * y[i] + z[i] * z[i];  // < 2^130
    }
    auto m3 = [&](int i, int j, int k) { return det3(x[i], y[i], z[i], x[j], y[j], z[j], x[k], y[k], z[k]); };  // < 2^195
    const Big det = -(l[1] * m3(0, 2, 3) + l[3] * m3(0, 1, 2) - l[0] * m3(1, 2, 3) - l[2] * m3(0, 1, 3));     // < 2^328
    return det.sign();
}

// ---------------------------------------------------------------- Delaunay triangulation with an infinite vertex
constexpr int INF = -1;

struct Cell {
    int v[4];   // finite cells: orient3d(v0, v1, v2, v3) > 0; infinite cells: positive once INF is replaced by any point beyond the hull facet
    int n[4];   // n[i]: the cell across the facet opposite v[i]
    bool alive;
};

struct Delaunay {
    const std::vector<P3> &pt;
    std::vector<Cell> cells;
    std::vector<int> free_list;
    int last = 0;  // a finite cell to start walks from

    explicit Delaunay(const std::vector<P3> &p) : pt(p) {}

    static int index_of(const Cell &c, int vertex)
    {
        for (int i = 0; i < 4; i++)
            if (c.v[i] == vertex) return i;
        return -1;
    }
    int inf_index(const Cell &c) const { return index_of(c, INF); }

    int new_cell(int a, int b, int c, int d)
    {
        int id;
        if (!free_list.empty()) {
            id = free_list.back();
            free_list.pop_back();
        } else {
            id = (int)cells.size();
            cells.push_back(Cell());
        }
        Cell &t = cells[id];
        t.v[0] = a, t.v[1] = b, t.v[2] = c, t.v[3] = d;
        t.n[0] = t.n[1] = t.n[2] = t.n[3] = -1;
        t.alive = true;
        return id;
    }

    // orientation of cell c with its vertex i replaced by the point q (c finite or infinite, v[i] may be INF; the others finite)
    int orient_replaced(const Cell &c, int i, int q) const
    {
        const P3 *p[4];
        for (int k = 0; k < 4; k++) p[k] = &pt[k == i ? q : c.v[k]];
        return orient3d(*p[0], *p[1], *p[2], *p[3]);
    }

    bool in_conflict(int ci, int q) const
    {
        const Cell &c = cells[ci];
        const int k = inf_index(c);
        if (k < 0) return insphere(pt[c.v[0]], pt[c.v[1]], pt[c.v[2]], pt[c.v[3]], pt[q]) > 0;
        const int o = orient_replaced(c, k, q);
        if (o != 0) return o > 0;  // strictly beyond the hull facet, or strictly on the inner side of its plane
        // in the plane of the hull facet: in conflict iff inside the facet's circumcircle = in conflict with the finite cell behind it
        const Cell &f = cells[c.n[k]];
        return insphere(pt[f.v[0]], pt[f.v[1]], pt[f.v[2]], pt[f.v[3]], pt[q]) > 0;
    }

    // the first four points that span space, as the first finite cell and its four infinite neighbours
    bool init(std::vector<int> &order)
    {
        const int n = (int)order.size();
        if (n < 4) return false;
        auto collinear = [&](const P3 &a, const P3 &b, const P3 &c) {
            const __int128 ux = (__int128)b.ix - a.ix, uy = (__int128)b.iy - a.iy, uz = (__int128)b.iz - a.iz;
            const __int128 vx = (__int128)c.ix - a.ix, vy = (__int128)c.iy - a.iy, vz = (__int128)c.iz - a.iz;
            const Big cx = Big(uy) * Big(vz) - Big(uz) * Big(vy), cy = Big(uz) * Big(vx) - Big(ux) * Big(vz), cz = Big(ux) * Big(vy) - Big(uy) * Big(vx);
            return cx.zero() && cy.zero() && cz.zero();
        };
        int i2 = -1, i3 = -1;
        for (int i = 2; i < n && i2 < 0; i++)
            if (!collinear(pt[order[0]], pt[order[1]], pt[order[i]])) i2 = i;
        if (i2 < 0) return false;
        for (int i = 2; i < n && i3 < 0; i++)
            if (i != i2 && orient3d(pt[order[0]], pt[order[1]], pt[order[i2]], pt[order[i]]) != 0) i3 = i;
        if (i3 < 0) return false;
        std::swap(order[2], order[i2]);
        if (i3 == 2) i3 = i2;  // (the point that sat at position 2 went to i2)
        std::swap(order[3], order[i3]);
        int a = order[0], b = order[1], c = order[2], d = order[3];
        if (orient3d(pt[a], pt[b], pt[c], pt[d]) < 0) std::swap(a, b);
        const int f = new_cell(a, b, c, d);
        // the infinite cell across the facet opposite vertex i: that vertex replaced by INF and two others swapped (INF lies on the other side)
        int inf[4];
        for (int i = 0; i < 4; i++) {
            int v[4] = {a, b, c, d};
            v[i] = INF;
            const int j = (i + 1) & 3, k = (i + 2) & 3;
            std::swap(v[j], v[k]);
            inf[i] = new_cell(v[0], v[1], v[2], v[3]);
        }
        // adjacency of the five cells by brute force: two cells are neighbours across the facets whose vertex sets are equal
        const int ids[5] = {f, inf[0], inf[1], inf[2], inf[3]};
        for (int x = 0; x < 5; x++)
            for (int i = 0; i < 4; i++) {
                Cell &cx = cells[ids[x]];
                for (int y = 0; y < 5; y++) {
                    if (x == y) continue;
                    const Cell &cy = cells[ids[y]];
                    for (int j = 0; j < 4; j++) {
                        bool same = true;
                        for (int k = 0; k < 4 && same; k++)
                            if (k != i && (index_of(cy, cx.v[k]) < 0 || index_of(cy, cx.v[k]) == j)) same = false;
                        if (same) cx.n[i] = ids[y];
                    }
                }
            }
        last = f;
        return true;
    }

    // a cell in conflict with point q: the finite cell that contains it, or an infinite cell whose hull facet it lies beyond
    int locate(int q)
    {
        int c = last;
        if (!cells[c].alive || inf_index(cells[c]) >= 0) {
            for (c = 0; c < (int)cells.size(); c++)
                if (cells[c].alive && inf_index(cells[c]) < 0) break;
        }
        int prev = -1;
        for (size_t steps = 0; steps < 4 * cells.size() + 64; steps++) {
            const Cell &t = cells[c];
            if (inf_index(t) >= 0) return c;  // walked out through a hull facet: q is beyond it (or in its plane, outside)
            int next = -1;
            for (int r = 0; r < 4; r++) {
                const int i = (r + (int)steps) & 3;  // vary the first facet tried: no cycling on degenerate inputs
                if (t.n[i] == prev) continue;
                if (orient_replaced(t, i, q) < 0) {
                    next = t.n[i];
                    break;
                }
            }
            if (next < 0) return c;
            prev = c;
            c = next;
        }
        // (never seen) fall back to a scan
        for (c = 0; c < (int)cells.size(); c++)
            if (cells[c].alive && in_conflict(c, q)) return c;
        return -1;
    }

    void insert(int q)
    {
        int start = locate(q);
        if (start < 0) return;
        if (!in_conflict(start, q)) {
            // on the boundary of the located cell (a facet, an edge): one of its neighbours is in conflict unless q is cospherical with everything around
            int found = -1;
            for (int i = 0; i < 4 && found < 0; i++)
                if (in_conflict(cells[start].n[i], q)) found = cells[start].n[i];
            if (found < 0) {
                for (int c = 0; c < (int)cells.size() && found < 0; c++)
                    if (cells[c].alive && in_conflict(c, q)) found = c;
            }
            if (found < 0) return;  // (cannot happen for a point distinct from the vertices)
            start = found;
        }
        // the cavity: the connected set of cells in conflict with q
        std::vector<int> cavity{start}, stack{start};
        std::unordered_map<int, char> state;  // 1 in conflict, 2 not
        state[start] = 1;
        struct BFacet { int cell, i, outside; };
        std::vector<BFacet> boundary;
        while (!stack.empty()) {
            const int c = stack.back();
            stack.pop_back();
            for (int i = 0; i < 4; i++) {
                const int nb = cells[c].n[i];
                auto it = state.find(nb);
                char s;
                if (it == state.end()) {
                    s = in_conflict(nb, q) ? 1 : 2;
                    state[nb] = s;
                    if (s == 1) {
                        cavity.push_back(nb);
                        stack.push_back(nb);
                    }
                } else {
                    s = it->second;
                }
                if (s == 2) boundary.push_back({c, i, nb});
            }
        }
        // one new cell per boundary facet: the cavity cell with the vertex opposite the facet replaced by q (same orientation: q is on that vertex's side)
        std::map<std::pair<int, int>, std::pair<int, int>> open_edges;  // edge of a boundary facet -> (new cell, local facet index) waiting for its twin
        std::vector<int> created;
        created.reserve(boundary.size());
        for (const BFacet &b : boundary) {
            const Cell old = cells[b.cell];
            int v[4] = {old.v[0], old.v[1], old.v[2], old.v[3]};
            v[b.i] = q;
            const int t = new_cell(v[0], v[1], v[2], v[3]);
            created.push_back(t);
            cells[t].n[b.i] = b.outside;
            Cell &out = cells[b.outside];
            for (int j = 0; j < 4; j++)
                if (out.n[j] == b.cell) {
                    // (a cell can touch the cavity through more than one facet: take the one whose vertices are this facet's)
                    bool same = true;
                    for (int k = 0; k < 4 && same; k++)
                        if (k != b.i && (index_of(out, old.v[k]) < 0 || index_of(out, old.v[k]) == j)) same = false;
                    if (same) out.n[j] = t;
                }
            for (int k = 0; k < 4; k++) {
                if (k == b.i) continue;
                // the facet opposite v[k] holds q and the two other vertices of the boundary facet
                int e[2], m = 0;
                for (int r = 0; r < 4; r++)
                    if (r != k && r != b.i) e[m++] = v[r];
                const std::pair<int, int> key(std::min(e[0], e[1]), std::max(e[0], e[1]));
                auto it = open_edges.find(key);
                if (it == open_edges.end()) {
                    open_edges[key] = {t, k};
                } else {
                    cells[t].n[k] = it->second.first;
                    cells[it->second.first].n[it->second.second] = t;
                    open_edges.erase(it);
                }
            }
        }
        for (int c : cavity) {
            cells[c].alive = false;
            free_list.push_back(c);
        }
        for (int t : created)
            if (inf_index(cells[t]) < 0) {
                last = t;
                break;
            }
    }
};

// Morton order of the points: consecutive insertions are close, so the walk in locate() is short
std::vector<int> spatial_order(const std::vector<P3> &pt)
{
    const int n = (int)pt.size();
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (const P3 &p : pt) {
        const double c[3] = {p.x, p.y, p.z};
        for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], c[k]), hi[k] = std::max(hi[k], c[k]);
    }
    std::vector<std::pair<uint64_t, int>> key(n);
    for (int i = 0; i < n; i++) {
        const double c[3] = {pt[i].x, pt[i].y, pt[i].z};
        uint64_t code = 0;
        uint32_t q[3];
        for (int k = 0; k < 3; k++) q[k] = hi[k] > lo[k] ? (uint32_t)std::min(1048575.0, (c[k] - lo[k]) / (hi[k] - lo[k]) * 1048576.0) : 0u;
        for (int b = 19; b >= 0; b--)
            for (int k = 0; k < 3; k++) code = (code << 1) | ((q[k] >> b) & 1u);
        key[i] = {code, i};
    }
    std::sort(key.begin(), key.end());
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = key[i].second;
    return order;
}

// CGAL's squared_radiusC3 (four points), in double
double squared_radius(const P3 &p, const P3 &q, const P3 &r, const P3 &s)
{
    const double qpx = q.x - p.x, qpy = q.y - p.y, qpz = q.z - p.z, qp2 = qpx * qpx + qpy * qpy + qpz * qpz;
    const double rpx = r.x - p.x, rpy = r.y - p.y, rpz = r.z - p.z, rp2 = rpx * rpx + rpy * rpy + rpz * rpz;
    const double spx = s.x - p.x, spy = s.y - p.y, spz = s.z - p.z, sp2 = spx * spx + spy * spy + spz * spz;
    auto det = [](double a00, double a01, double a02, double a10, double a11, double a12, double a20, double a21, double a22) {
        return a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
    };
    const double num_x = det(qpy, qpz, qp2, rpy, rpz, rp2, spy, spz, sp2);
    const double num_y = det(qpx, qpz, qp2, rpx, rpz, rp2, spx, spz, sp2);
    const double num_z = det(qpx, qpy, qp2, rpx, rpy, rp2, spx, spy, sp2);
    const double den = det(qpx, qpy, qpz, rpx, rpy, rpz, spx, spy, spz);
    return (num_x * num_x + num_y * num_y + num_z * num_z) / (4.0 * den * den);
}

// float coordinates -> integers on one common grid 2^g (g = the smallest exponent any coordinate needs, but never more than 61 bits
// below the largest magnitude: coordinates 2^-38 times smaller than the extent of the data are rounded to that grid)
void integer_coordinates(std::vector<P3> &pt)
{
    int emax = -100000, emin = 100000;
    auto scan = [&](double v) {
        if (v == 0.0) return;
        int e;
        std::frexp(v, &e);            // |v| = m 2^e, m in [0.5, 1): a float needs bits 2^(e-1) .. 2^(e-24)
        emax = std::max(emax, e);
        emin = std::min(emin, e - 24);
    };
    for (const P3 &p : pt) scan(p.x), scan(p.y), scan(p.z);
    if (emax < emin) emax = emin = 0;  // all zero
    const int g = std::max(emin, emax - 61);
    for (P3 &p : pt) {
        p.ix = (int64_t)std::llround(std::ldexp(p.x, -g));
        p.iy = (int64_t)std::llround(std::ldexp(p.y, -g));
        p.iz = (int64_t)std::llround(std::ldexp(p.z, -g));
    }
}

struct AlphaResult {
    std::vector<int32_t> faces;   // 3 per face, indices of the input rows
    std::vector<int32_t> cells;   // 4 per finite Delaunay cell (input row indices), for tests
    double alpha = 0.0;
    int components = 0;
};

AlphaResult alpha_shape(const float *rows, int nrows, int cols, double forced_alpha)
{
    AlphaResult res;
    // distinct points; a later row with the same coordinates takes the index over (alpha_shapes.cpp:49,56: vertex_indices[p] = i)
    std::vector<P3> pt;
    std::vector<int> row_of;
    {
        std::map<std::array<double, 3>, int> seen;
        for (int i = 0; i < nrows; i++) {
            const float *r = rows + (size_t)i * cols;
            std::array<double, 3> c;
            if (cols == 3)
                c = {r[0], r[1], r[2]};
            else
                c = {(double)(float)(r[0] / r[3]), (double)(float)(r[1] / r[3]), (double)(float)(r[2] / r[3])};  // float division, as in the reference
            auto it = seen.find(c);
            if (it == seen.end()) {
                seen[c] = (int)pt.size();
                P3 p;
                p.x = c[0], p.y = c[1], p.z = c[2];
                p.ix = p.iy = p.iz = 0;
                pt.push_back(p);
                row_of.push_back(i);
            } else {
                row_of[it->second] = i;
            }
        }
    }
    integer_coordinates(pt);
    Delaunay dt(pt);
    std::vector<int> order = spatial_order(pt);
    if (!dt.init(order)) return res;  // fewer than four points, or all in one plane: no cells, no faces (the reference asserts one solid component)
    for (size_t i = 4; i < order.size(); i++) dt.insert(order[i]);

    // alpha of the finite cells
    const int nc = (int)dt.cells.size();
    std::vector<double> calpha(nc, 0.0);
    std::vector<char> finite(nc, 0);
    for (int c = 0; c < nc; c++) {
        const Cell &t = dt.cells[c];
        if (!t.alive || dt.inf_index(t) >= 0) continue;
        finite[c] = 1;
        calpha[c] = squared_radius(pt[t.v[0]], pt[t.v[1]], pt[t.v[2]], pt[t.v[3]]);
        for (int k = 0; k < 4; k++) res.cells.push_back(row_of[t.v[k]]);
    }
    // find_alpha_solid
    std::vector<double> vmin(pt.size(), -1.0);
    for (int c = 0; c < nc; c++)
        if (finite[c])
            for (int k = 0; k < 4; k++) {
                double &m = vmin[dt.cells[c].v[k]];
                m = m < 0.0 ? calpha[c] : std::min(m, calpha[c]);
            }
    double alpha_solid = 0.0;
    for (double m : vmin) alpha_solid = std::max(alpha_solid, m);
    // the spectrum (REGULARIZED mode: cells only; distinct, positive, ascending)
    std::vector<double> spectrum;
    for (int c = 0; c < nc; c++)
        if (finite[c] && calpha[c] > 0.0) spectrum.push_back(calpha[c]);
    std::sort(spectrum.begin(), spectrum.end());
    spectrum.erase(std::unique(spectrum.begin(), spectrum.end()), spectrum.end());
    if (spectrum.empty()) return res;
    auto components = [&](double alpha) {
        std::vector<char> seen(nc, 0);
        int count = 0;
        std::vector<int> stack;
        for (int c = 0; c < nc; c++) {
            if (!finite[c] || seen[c] || calpha[c] > alpha) continue;
            count++;
            seen[c] = 1;
            stack.push_back(c);
            while (!stack.empty()) {
                const int x = stack.back();
                stack.pop_back();
                for (int k = 0; k < 4; k++) {
                    const int y = dt.cells[x].n[k];
                    if (finite[y] && !seen[y] && calpha[y] <= alpha) {
                        seen[y] = 1;
                        stack.push_back(y);
                    }
                }
            }
        }
        return count;
    };
    // find_optimal_alpha(1)
    size_t first = std::lower_bound(spectrum.begin(), spectrum.end(), alpha_solid) - spectrum.begin();
    if (first >= spectrum.size()) first = spectrum.size() - 1;
    size_t opt;
    if (components(alpha_solid) == 1) {
        opt = first + 1 < spectrum.size() ? first + 1 : first;
    } else {
        ptrdiff_t len = (ptrdiff_t)spectrum.size() - (ptrdiff_t)first - 1;
        while (len > 0) {
            const ptrdiff_t half = len / 2;
            const size_t middle = first + half;
            if (components(spectrum[middle]) > 1) {
                first = middle + 1;
                len = len - half - 1;
            } else {
                len = half;
            }
        }
        opt = first + 1 < spectrum.size() ? first + 1 : first;
    }
    res.alpha = spectrum[opt];
    const double alpha = forced_alpha > 0.0 ? forced_alpha : res.alpha;
    res.components = components(alpha);
    // REGULAR facets, normals out of the solid
    for (int c = 0; c < nc; c++) {
        if (!finite[c] || calpha[c] > alpha) continue;  // c interior
        const Cell &t = dt.cells[c];
        for (int i = 0; i < 4; i++) {
            const int nb = t.n[i];
            if (finite[nb] && calpha[nb] <= alpha) continue;  // both interior: an INTERIOR facet
            int f[3], m = 0;
            for (int k = 0; k < 4; k++)
                if (k != i) f[m++] = t.v[k];
            // ((f1 - f0) x (f2 - f0)) . (v_i - f0) must be negative: the interior cell's fourth vertex is behind the face
            if (orient3d(pt[f[0]], pt[f[1]], pt[f[2]], pt[t.v[i]]) > 0) std::swap(f[1], f[2]);
            for (int k = 0; k < 3; k++) res.faces.push_back(row_of[f[k]]);
        }
    }
    return res;
}

}  // namespace

// C entry points (ctypes in tests/test_meshing_cpu.py; a C caller): points = rows x cols floats (cols 3: Cartesian, 4: homogeneous).
// faces / cells may be null (count only).  Returns 0, 1 when a buffer is too small (the counts say what is needed), 2 for a bad argument,
// 3 when memory ran out.
extern "C" int mvs_alpha_shape_faces(const float *points, int rows, int cols, float forced_alpha, int32_t *faces, int face_capacity, int *face_count,
                                     float *alpha, int *solid_components)
{
    if (!points || rows < 0 || (cols != 3 && cols != 4) || !face_count) return 2;
    AlphaResult r;
    try {
        r = alpha_shape(points, rows, cols, forced_alpha);
    } catch (...) {  // (allocation failure: no exception may cross the C boundary)
        return 3;
    }
    *face_count = (int)(r.faces.size() / 3);
    if (alpha) *alpha = (float)r.alpha;
    if (solid_components) *solid_components = r.components;
    if (faces) {
        if (face_capacity < *face_count) return 1;
        if (!r.faces.empty()) std::memcpy(faces, r.faces.data(), r.faces.size() * sizeof(int32_t));  // (memcpy from a null source is undefined even for 0 bytes: UBSan)
    }
    return 0;
}

extern "C" int mvs_delaunay3_cells(const float *points, int rows, int cols, int32_t *cells, int cell_capacity, int *cell_count)
{
    if (!points || rows < 0 || (cols != 3 && cols != 4) || !cell_count) return 2;
    AlphaResult r;
    try {
        r = alpha_shape(points, rows, cols, 0.0);
    } catch (...) {
        return 3;
    }
    *cell_count = (int)(r.cells.size() / 4);
    if (cells) {
        if (cell_capacity < *cell_count) return 1;
        if (!r.cells.empty()) std::memcpy(cells, r.cells.data(), r.cells.size() * sizeof(int32_t));
    }
    return 0;
}

// recon.hpp:33-34
Mat alphaShapeFaces(const Mat points, float *alpha)
{
    if (points.rows == 0 || points.cols == 0) return Mat(0, 3, mvs::S32C1);  // alpha_shapes.cpp:38-39
    if (points.cols != 3 && points.cols != 4) throw std::runtime_error("alphaShapeFaces: points must have 3 or 4 columns");  // alpha_shapes.cpp:60: assert(false)
    const AlphaResult r = alpha_shape(points.ptr<float>(), points.rows, points.cols, 0.0);
    if (r.components != 1 && !r.faces.empty()) throw std::runtime_error("alphaShapeFaces: the alpha shape is not one solid component");  // alpha_shapes.cpp:75
    if (alpha) *alpha = (float)r.alpha;
    Mat faces((int)(r.faces.size() / 3), 3, mvs::S32C1);
    if (!r.faces.empty()) std::memcpy(faces.ptr<int32_t>(), r.faces.data(), r.faces.size() * sizeof(int32_t));
    return faces;
}

Mat alphaShapeFaces(const Mat points) { return alphaShapeFaces(points, nullptr); }  // alpha_shapes.cpp:101-104
