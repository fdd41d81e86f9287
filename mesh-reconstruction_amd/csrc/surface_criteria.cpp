// surface_criteria.cpp -- the facet criteria of the reference's surface mesher as a pass over the surface-nets mesh.
//
// cgal_poisson.cpp:50-52, 95-102 hands CGAL::make_surface_mesh three numbers: a LOWER bound on the facets' angles (sm_angle = 20 degrees),
// an UPPER bound on the radius of their surface Delaunay balls (sm_radius = 300 average spacings: in practice never binding) and an upper
// bound on the distance between a facet and the surface (sm_distance = 0.375 average spacings).  CGAL meets them by Delaunay refinement:
// it inserts surface points until no facet is "bad".  The grid mesher of csrc/poisson.hip meets the distance bound by construction (node
// spacing <= 0.75 average spacings, header of that file) and the radius bound trivially (its facets span at most two cells), but surface
// nets put a vertex wherever the mean edge crossing of a cell falls, so two vertices of a quad can lie arbitrarily close together: a
// few per cent of its triangles are needles or caps.  This pass removes them with the two local operations that do not move a vertex off
// the level set (every vertex that survives is one the mesher placed):
//   * edge collapse  u -> v  (u disappears; the two facets on the edge go, the others around u now end in v), only when the edge has
//     exactly two facets and the link condition holds (the only common neighbours of u and v are the two opposite vertices: the surface
//     stays a manifold wherever it was one), no surviving facet turns over, and u stays within a quarter of the distance bound of the
//     new facets' planes;
//   * edge flip (the other diagonal of the quad formed by the edge's two facets), only when that diagonal is not an edge already, the
//     two new facets keep the orientation, and the two diagonals pass within a quarter of the distance bound of each other (the quad is
//     nearly flat).
// A facet below the angle bound is offered all nine candidates on its three edges (collapse either way, flip); the valid candidate that
// leaves the best worst-angle among the facets it touches is applied if that is a strict improvement of their worst angle before.  Work
// list in facet order, then first in first out; every facet an operation touches and leaves below the bound is queued again.  Purely
// sequential host code, deterministic; like the reference's mesher (CGAL on the CPU) it is not on the hot path.
// What it reports is what the criteria ask: how many facets are still below the angle bound (0 on every surface of the test-suite),
// how many are above the radius bound, the smallest angle and the largest circumradius found.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mvs.h"
#include "surface_internal.hpp"

namespace {

struct V3 {
    double x, y, z;
};
inline V3 sub(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// quality of a triangle: minus the largest cosine of its angles (-1: degenerate, -0.5: equilateral); min angle >= theta  <=>  q >= -cos(theta)
double quality(const V3 &a, const V3 &b, const V3 &c)
{
    const V3 ab = sub(b, a), bc = sub(c, b), ca = sub(a, c);
    const double lab = dot(ab, ab), lbc = dot(bc, bc), lca = dot(ca, ca);
    if (!(lab > 0.0) || !(lbc > 0.0) || !(lca > 0.0)) return -1.0;
    const double cA = -dot(ab, ca) / std::sqrt(lab * lca), cB = -dot(ab, bc) / std::sqrt(lab * lbc), cC = -dot(bc, ca) / std::sqrt(lbc * lca);
    return -std::max(cA, std::max(cB, cC));
}

struct Work {
    std::vector<V3> p;
    std::vector<int32_t> f;             // 3 per facet; f[3 i] < 0: removed
    std::vector<std::vector<int>> inc;  // facets around a vertex
    std::vector<long long> stamp;  // (two stamps per collapse candidate: 64 bits, so that no mesh size can wrap them)
    long long stamp_now = 0;
    double q_bound = 0.0, guard = 0.0;
    double last_nearest = 0.0;  // of the last try_collapse that returned a result: distance of the vanishing vertex from the nearest new facet's plane
    int collapses = 0, flips = 0;

    // undo log of the rescue round (below): every facet and incidence list is saved before it changes; rolled back newest first
    struct Journal {
        bool on = false;
        std::vector<std::pair<int, std::array<int32_t, 3>>> faces;
        std::vector<std::pair<int, std::vector<int>>> lists;
        int collapses = 0, flips = 0;
    } journal;
    void save_face(int face)
    {
        if (journal.on) journal.faces.push_back({face, {f[3 * face], f[3 * face + 1], f[3 * face + 2]}});
    }
    void save_list(int v)
    {
        if (journal.on) journal.lists.push_back({v, inc[(size_t)v]});
    }
    void journal_begin()
    {
        journal.on = true;
        journal.faces.clear();
        journal.lists.clear();
        journal.collapses = collapses;
        journal.flips = flips;
    }
    void journal_commit() { journal.on = false; }
    void journal_rollback()
    {
        for (size_t i = journal.faces.size(); i-- > 0;)
            for (int k = 0; k < 3; k++) f[3 * journal.faces[i].first + k] = journal.faces[i].second[(size_t)k];
        for (size_t i = journal.lists.size(); i-- > 0;) inc[(size_t)journal.lists[i].first] = journal.lists[i].second;
        collapses = journal.collapses;
        flips = journal.flips;
        journal.on = false;
    }

    double q_of(int face) const { return quality(p[f[3 * face]], p[f[3 * face + 1]], p[f[3 * face + 2]]); }
    V3 normal_of(int a, int b, int c) const { return cross(sub(p[b], p[a]), sub(p[c], p[a])); }
    bool has(int face, int v) const { return f[3 * face] == v || f[3 * face + 1] == v || f[3 * face + 2] == v; }
    int third(int face, int u, int v) const
    {
        for (int k = 0; k < 3; k++)
            if (f[3 * face + k] != u && f[3 * face + k] != v) return f[3 * face + k];
        return -1;
    }
    // does `face` run u -> v along its boundary?
    bool directed(int face, int u, int v) const
    {
        for (int k = 0; k < 3; k++)
            if (f[3 * face + k] == u && f[3 * face + (k + 1) % 3] == v) return true;
        return false;
    }
    int edge_facets(int u, int v, int out[2]) const
    {
        int n = 0;
        for (int face : inc[u])
            if (has(face, v)) {
                if (n < 2) out[n] = face;
                n++;
            }
        return n;
    }
    static void forget(std::vector<int> &list, int face)
    {
        const auto it = std::find(list.begin(), list.end(), face);
        if (it != list.end()) list.erase(it);
    }
    bool connected(int a, int b) const
    {
        for (int face : inc[a])
            if (has(face, b)) return true;
        return false;
    }

    // ---- collapse u -> v ----
    struct Candidate {
        int kind = 0;  // 1 collapse, 2 flip
        int u = 0, v = 0;
        double after = -2.0;
    };

    double try_collapse(int u, int v, double *before_out)
    {
        int e[2];
        if (edge_facets(u, v, e) != 2) return -2.0;
        const int c = third(e[0], u, v), d = third(e[1], u, v);
        if (c < 0 || d < 0 || c == d) return -2.0;
        if (inc[c].size() <= 3 || inc[d].size() <= 3 || inc[u].size() + inc[v].size() < 7) return -2.0;
        // link condition: neighbours of u that are neighbours of v = {c, d}
        stamp_now++;
        for (int face : inc[v])
            for (int k = 0; k < 3; k++) stamp[f[3 * face + k]] = stamp_now;
        int shared = 0;
        const long long seen = ++stamp_now;  // (v's marks are stamp_now - 1 from here on)
        for (int face : inc[u])
            for (int k = 0; k < 3; k++) {
                const int w = f[3 * face + k];
                if (w == u || w == v) continue;
                if (stamp[w] == seen - 1) {
                    shared++;
                    stamp[w] = seen;  // count a neighbour once
                }
            }
        if (shared != 2) return -2.0;
        double before = 0.0, after = 0.0, nearest = 1e300;
        bool any = false;
        for (int face : inc[u]) {
            const double q0 = q_of(face);
            before = std::min(before, q0);
            if (face == e[0] || face == e[1]) continue;
            int t[3];
            for (int k = 0; k < 3; k++) t[k] = f[3 * face + k] == u ? v : f[3 * face + k];
            const double q1 = quality(p[t[0]], p[t[1]], p[t[2]]);
            after = std::min(after, q1);
            const V3 n0 = normal_of(f[3 * face], f[3 * face + 1], f[3 * face + 2]), n1 = normal_of(t[0], t[1], t[2]);
            const double l0 = dot(n0, n0), l1 = dot(n1, n1);
            if (!(l1 > 0.0)) return -2.0;
            if (q0 > -0.9962 && dot(n0, n1) < 0.3 * std::sqrt(l0 * l1)) return -2.0;  // (a facet below 5 degrees has no normal worth keeping)
            nearest = std::min(nearest, std::fabs(dot(n1, sub(p[u], p[t[0]]))) / std::sqrt(l1));
            any = true;
        }
        if (!any || nearest > guard) return -2.0;
        *before_out = before;
        last_nearest = nearest;
        return after;
    }

    void do_collapse(int u, int v, std::vector<int> &touched)
    {
        int e[2];
        edge_facets(u, v, e);
        for (int k = 0; k < 2; k++) {
            const int face = e[k];
            for (int j = 0; j < 3; j++) {
                const int w = f[3 * face + j];
                if (w == u) continue;
                save_list(w);
                forget(inc[w], face);
            }
            save_face(face);
            f[3 * face] = f[3 * face + 1] = f[3 * face + 2] = -1;
        }
        save_list(v);
        save_list(u);
        for (int face : inc[u]) {
            if (face == e[0] || face == e[1]) continue;
            save_face(face);
            for (int j = 0; j < 3; j++)
                if (f[3 * face + j] == u) f[3 * face + j] = v;
            inc[v].push_back(face);
            touched.push_back(face);
        }
        inc[u].clear();
        collapses++;
    }

    // ---- flip the edge u - v ----
    double try_flip(int u, int v, double *before_out)
    {
        int e[2];
        if (edge_facets(u, v, e) != 2) return -2.0;
        int f1 = e[0], f2 = e[1];
        if (!directed(f1, u, v)) std::swap(f1, f2);
        if (!directed(f1, u, v) || !directed(f2, v, u)) return -2.0;  // not consistently oriented here
        const int c = third(f1, u, v), d = third(f2, u, v);
        if (c < 0 || d < 0 || c == d || connected(c, d)) return -2.0;
        if (inc[u].size() <= 3 || inc[v].size() <= 3) return -2.0;
        const double before = std::min(q_of(f1), q_of(f2));
        const double after = std::min(quality(p[u], p[d], p[c]), quality(p[d], p[v], p[c]));
        const V3 n1 = normal_of(u, v, c), n2 = normal_of(v, u, d);
        const V3 ref = {n1.x + n2.x, n1.y + n2.y, n1.z + n2.z};
        const V3 m1 = normal_of(u, d, c), m2 = normal_of(d, v, c);
        const double lr = dot(ref, ref), l1 = dot(m1, m1), l2 = dot(m2, m2);
        if (!(lr > 0.0) || !(l1 > 0.0) || !(l2 > 0.0)) return -2.0;
        if (dot(m1, ref) < 0.3 * std::sqrt(l1 * lr) || dot(m2, ref) < 0.3 * std::sqrt(l2 * lr)) return -2.0;
        // distance between the two diagonals (as lines): how far the surface moves when one replaces the other
        const V3 x = cross(sub(p[v], p[u]), sub(p[d], p[c]));
        const double lx = dot(x, x);
        if (!(lx > 0.0) || std::fabs(dot(x, sub(p[c], p[u]))) / std::sqrt(lx) > guard) return -2.0;
        *before_out = before;
        return after;
    }

    void do_flip(int u, int v, std::vector<int> &touched)
    {
        int e[2];
        edge_facets(u, v, e);
        int f1 = e[0], f2 = e[1];
        if (!directed(f1, u, v)) std::swap(f1, f2);
        const int c = third(f1, u, v), d = third(f2, u, v);
        save_face(f1);
        save_face(f2);
        save_list(u);
        save_list(v);
        save_list(c);
        save_list(d);
        f[3 * f1] = u, f[3 * f1 + 1] = d, f[3 * f1 + 2] = c;
        f[3 * f2] = d, f[3 * f2 + 1] = v, f[3 * f2 + 2] = c;
        forget(inc[v], f1);
        inc[d].push_back(f1);
        forget(inc[u], f2);
        inc[c].push_back(f2);
        touched.push_back(f1);
        touched.push_back(f2);
        flips++;
    }

    bool improve(int face, std::vector<int> &touched)
    {
        Candidate best;
        for (int k = 0; k < 3; k++) {
            const int u = f[3 * face + k], v = f[3 * face + (k + 1) % 3];
            double before = 0.0;
            double a = try_collapse(u, v, &before);
            if (a > before + 1e-12 && a > best.after) best = {1, u, v, a};
            a = try_collapse(v, u, &before);
            if (a > before + 1e-12 && a > best.after) best = {1, v, u, a};
            a = try_flip(u, v, &before);
            if (a > before + 1e-12 && a > best.after) best = {2, u, v, a};
        }
        if (best.kind == 1) do_collapse(best.u, best.v, touched);
        else if (best.kind == 2) do_flip(best.u, best.v, touched);
        return best.kind != 0;
    }

    // The rescue round (round 5): a facet every single operation would leave next to a WORSE facet -- a local minimum of the greedy rule
    // above; one or two per 60 000 facets on the test surfaces -- gets a look-ahead: each valid operation on its edges (best result
    // first), improvement or not, is applied on trial, the facets it leaves below the bound are handed to improve() in turn (at most
    // RESCUE_STEPS follow-up operations), and the sequence is kept only if NO facet it touched ends below the bound; otherwise every
    // change is rolled back.  Same guards as everywhere: no vertex moves, none is added, the link condition and the distance bound hold
    // for every single operation.
    static constexpr int RESCUE_STEPS = 6;
    bool alive_bad(int face) const { return f[3 * face] >= 0 && q_of(face) < q_bound; }
    bool rescue(int face, std::vector<int> &touched)
    {
        std::vector<Candidate> cands;
        for (int k = 0; k < 3; k++) {
            const int u = f[3 * face + k], v = f[3 * face + (k + 1) % 3];
            double before = 0.0;
            double a = try_collapse(u, v, &before);
            if (a > -2.0) cands.push_back({1, u, v, a});
            a = try_collapse(v, u, &before);
            if (a > -2.0) cands.push_back({1, v, u, a});
            a = try_flip(u, v, &before);
            if (a > -2.0) cands.push_back({2, u, v, a});
        }
        std::stable_sort(cands.begin(), cands.end(), [](const Candidate &x, const Candidate &y) { return x.after > y.after; });
        for (const Candidate &c : cands) {
            journal_begin();
            std::vector<int> trial, work;
            if (c.kind == 1) do_collapse(c.u, c.v, trial);
            else do_flip(c.u, c.v, trial);
            for (int t : trial)
                if (alive_bad(t)) work.push_back(t);
            bool ok = true;
            int steps = 0;
            for (size_t head = 0; head < work.size(); head++) {
                const int j = work[head];
                if (!alive_bad(j)) continue;
                if (steps++ >= RESCUE_STEPS) {
                    ok = false;
                    break;
                }
                std::vector<int> t2;
                if (!improve(j, t2)) {
                    ok = false;
                    break;
                }
                for (int t : t2) {
                    trial.push_back(t);
                    if (alive_bad(t)) work.push_back(t);
                }
            }
            for (size_t i = 0; ok && i < trial.size(); i++)
                if (alive_bad(trial[i])) ok = false;
            if (ok && alive_bad(face)) ok = false;
            if (ok) {
                journal_commit();
                touched.insert(touched.end(), trial.begin(), trial.end());
                return true;
            }
            journal_rollback();
        }
        return false;
    }
};

}  // namespace

extern "C" int mvs_surface_from_mesh(const float *vertices, int vertex_count, const int32_t *faces, int face_count, float average_spacing, mvs_surface **out)
{
    if (!out || vertex_count < 0 || face_count < 0 || (vertex_count && !vertices) || (face_count && !faces) || !(average_spacing >= 0.0f)) return MVS_EINVAL;
    *out = nullptr;
    for (int i = 0; i < 3 * face_count; i++)
        if (faces[i] < 0 || faces[i] >= vertex_count) return MVS_EINVAL;
    mvs_surface *s = new (std::nothrow) mvs_surface;
    if (!s) return MVS_ENOMEM;
    try {
        s->vertices.assign(vertices, vertices + 4 * (size_t)vertex_count);
        s->faces.assign(faces, faces + 3 * (size_t)face_count);
    } catch (...) {
        delete s;
        return MVS_ENOMEM;
    }
    s->spacing = average_spacing;
    *out = s;
    return MVS_OK;
}

extern "C" int mvs_surface_enforce_criteria(mvs_surface *s, float min_angle_deg, float max_radius, float max_distance, mvs_criteria_report *report)
{
    if (!s || !(min_angle_deg >= 0.0f) || !(min_angle_deg < 60.0f) || !(max_radius > 0.0f) || !(max_distance >= 0.0f)) return MVS_EINVAL;
    const int nv = (int)(s->vertices.size() / 4), nf = (int)(s->faces.size() / 3);
    mvs_criteria_report r;
    std::memset(&r, 0, sizeof r);
    r.min_angle_deg = 180.0f;
    try {
        Work w;
        w.p.resize((size_t)nv);
        for (int i = 0; i < nv; i++) w.p[(size_t)i] = {(double)s->vertices[4 * (size_t)i], (double)s->vertices[4 * (size_t)i + 1], (double)s->vertices[4 * (size_t)i + 2]};
        w.f = s->faces;
        w.inc.resize((size_t)nv);
        w.stamp.assign((size_t)nv, 0);
        for (int i = 0; i < nf; i++) {
            const int a = w.f[3 * i], b = w.f[3 * i + 1], c = w.f[3 * i + 2];
            if (a == b || b == c || a == c) {  // not a triangle: drop it
                w.f[3 * i] = w.f[3 * i + 1] = w.f[3 * i + 2] = -1;
                continue;
            }
            w.inc[(size_t)a].push_back(i), w.inc[(size_t)b].push_back(i), w.inc[(size_t)c].push_back(i);
        }
        w.q_bound = -std::cos((double)min_angle_deg * 3.14159265358979323846 / 180.0);
        w.guard = 0.25 * (double)max_distance;
        std::vector<int> queue, touched;
        for (int i = 0; i < nf; i++)
            if (w.f[3 * i] >= 0 && w.q_of(i) < w.q_bound) queue.push_back(i);
        const size_t budget = 2 * (size_t)nf + 1000;  // (collapses are finite by themselves; the cap is for flips chasing each other: a clean surface needs nf / 50 operations)
        size_t done = 0;
        // second round: the facets the first one could not help, with half of the distance bound to move in instead of a quarter; third
        // round (round 5): with the whole bound -- the distance the reference itself allows a facet (cgal_poisson.cpp:52)
        for (int round = 0; round < 3; round++) {
            for (size_t head = 0; head < queue.size() && done < budget; head++) {
                const int face = queue[head];
                if (w.f[3 * face] < 0 || w.q_of(face) >= w.q_bound) continue;
                touched.clear();
                if (w.improve(face, touched)) {
                    done++;
                    for (int t : touched)
                        if (w.f[3 * t] >= 0 && w.q_of(t) < w.q_bound) queue.push_back(t);
                }
            }
            std::vector<int> left;
            for (int face : queue)
                if (w.f[3 * face] >= 0 && w.q_of(face) < w.q_bound) left.push_back(face);
            std::sort(left.begin(), left.end());
            left.erase(std::unique(left.begin(), left.end()), left.end());
            if (left.empty()) break;
            queue.swap(left);
            w.guard = (round == 0 ? 0.5 : 1.0) * (double)max_distance;
        }
        // rescue round: what three greedy rounds left below the bound, with look-ahead (Work::rescue), under the whole distance bound
        {
            std::vector<int> left;
            for (int face : queue)
                if (w.alive_bad(face)) left.push_back(face);
            std::sort(left.begin(), left.end());
            left.erase(std::unique(left.begin(), left.end()), left.end());
            w.guard = (double)max_distance;
            for (size_t head = 0; head < left.size() && head < 4096; head++) {  // (a clean surface leaves a handful; the cap bounds a hopeless one)
                const int face = left[head];
                if (!w.alive_bad(face)) continue;
                touched.clear();
                (void)w.rescue(face, touched);
            }
        }
        // border trim: a facet that is still below the bound and lies on the sheet's OUTLINE is a defect of where the samples' support cut the
        // level set (csrc/poisson.hip step 4a), not of the surface: it is removed when that only moves the outline -- all three or two of
        // its edges are border edges (an isolated facet, an ear), or one is and the opposite vertex is not on the border (else the outline
        // would pass through that vertex twice); its other edges must have exactly two facets.  No other facet changes, so the candidates are the facets below the bound now; passes until
        // none goes (a facet behind an ear becomes an ear).  On the pipeline's own cloud 53 of the 56 facets the rounds above leave are such.
        {
            std::vector<int> bad;
            for (int i = 0; i < nf; i++)
                if (w.alive_bad(i)) bad.push_back(i);
            auto facets_on = [&](int a, int b) {
                int two[2];
                return w.edge_facets(a, b, two);
            };
            auto border_edge = [&](int a, int b) { return facets_on(a, b) == 1; };
            auto border_vertex = [&](int v) {
                for (int face : w.inc[(size_t)v])
                    for (int k = 0; k < 3; k++)
                        if (w.f[3 * face + k] != v && border_edge(v, w.f[3 * face + k])) return true;
                return false;
            };
            for (bool changed = true; changed;) {
                changed = false;
                for (int face : bad) {
                    if (w.f[3 * face] < 0) continue;
                    const int a = w.f[3 * face], b = w.f[3 * face + 1], c = w.f[3 * face + 2];
                    const int on[3] = {facets_on(a, b), facets_on(b, c), facets_on(c, a)};
                    const bool eb[3] = {on[0] == 1, on[1] == 1, on[2] == 1};
                    const int nb = (int)eb[0] + (int)eb[1] + (int)eb[2];
                    if (nb == 0 || on[0] > 2 || on[1] > 2 || on[2] > 2) continue;  // (an edge with more than two facets: nothing is defined there)
                    if (nb == 1 && border_vertex(eb[0] ? c : (eb[1] ? a : b))) continue;
                    Work::forget(w.inc[(size_t)a], face), Work::forget(w.inc[(size_t)b], face), Work::forget(w.inc[(size_t)c], face);
                    w.f[3 * face] = w.f[3 * face + 1] = w.f[3 * face + 2] = -1;
                    r.facets_trimmed++;
                    changed = true;
                }
            }
        }
        r.collapses = w.collapses, r.flips = w.flips;
        // compact: vertices still used, in their old order; facets in their old order
        std::vector<int> renum((size_t)nv, -1);
        for (int i = 0; i < nf; i++)
            if (w.f[3 * i] >= 0)
                for (int k = 0; k < 3; k++) renum[(size_t)w.f[3 * i + k]] = 0;
        int kept_v = 0;
        std::vector<float> vout;
        for (int i = 0; i < nv; i++)
            if (renum[(size_t)i] == 0) {
                renum[(size_t)i] = kept_v++;
                for (int k = 0; k < 4; k++) vout.push_back(s->vertices[4 * (size_t)i + k]);
            }
        std::vector<int32_t> fout;
        double min_q = 0.0, max_r2 = 0.0;
        for (int i = 0; i < nf; i++) {
            if (w.f[3 * i] < 0) continue;
            const V3 &a = w.p[(size_t)w.f[3 * i]], &b = w.p[(size_t)w.f[3 * i + 1]], &c = w.p[(size_t)w.f[3 * i + 2]];
            const double q = quality(a, b, c);
            if (q < w.q_bound) r.facets_below_angle++;
            min_q = std::min(min_q, q);
            // circumradius: |ab| |bc| |ca| / (4 area)
            const V3 n = cross(sub(b, a), sub(c, a));
            const double area2 = dot(n, n);  // (2 area)^2
            const double l = dot(sub(b, a), sub(b, a)) * dot(sub(c, b), sub(c, b)) * dot(sub(a, c), sub(a, c));
            const double r2 = area2 > 0.0 ? l / (4.0 * area2) : 1e300;
            if (r2 > (double)max_radius * (double)max_radius) r.facets_above_radius++;
            if (r2 < 1e299) max_r2 = std::max(max_r2, r2);
            for (int k = 0; k < 3; k++) fout.push_back(renum[(size_t)w.f[3 * i + k]]);
        }
        r.min_angle_deg = fout.empty() ? 180.0f : (float)(std::acos(std::min(1.0, std::max(-1.0, -min_q))) * 180.0 / 3.14159265358979323846);
        r.max_circumradius = (float)std::sqrt(max_r2);
        s->vertices.swap(vout);
        s->faces.swap(fout);
    } catch (...) {
        return MVS_ENOMEM;
    }
    if (report) *report = r;
    return MVS_OK;
}

// ---- simplification (round 5) ------------------------------------------------------------------------------------------------------
// cgal_poisson.cpp:95-97's criteria bound a facet from BELOW only in its angles; its size may reach 300 average spacings as long as it
// stays within 0.375 spacings of the surface -- CGAL's Delaunay refinement therefore returns a mesh whose density follows the curvature,
// while surface nets return the grid's (171 k vertices on the pipeline's cloud where a few thousand satisfy the criteria).  This pass
// removes vertices the criteria do not need, by the collapse of surface_criteria's first half and under its guards: an edge (shortest
// first; lengths never change, no vertex moves) is collapsed u -> v when the link condition holds, no surviving facet turns by more than
// ~72 degrees, EVERY facet the collapse rewrites keeps min_angle_deg, and the vertex that disappears stays within what is left of ITS
// distance budget of the new facets' planes: a vertex carries the largest distance any vertex merged into it was moved (err[v] =
// max(err[v], err[u] + d)), and err[u] + d <= max_distance.  That bounds where the VERTICES of the old surface end up; what a large facet does
// between its corners -- the chord cutting across a curved surface -- is bounded by the surface's own normals: every vertex keeps the
// normal the input mesh had there (area-weighted mean of its facets' normals), and a rewritten facet T with longest edge L whose normal
// makes at most the angle theta with the kept normals of its three corners may lie (L / 2) tan(theta / 2) inside the surface (the sagitta of
// the circular arc those normals describe); that, plus the largest err of its corners, must stay within max_distance too.  Vertices on a border or on
// an edge that does not have exactly two facets are never removed (the sheet's outline and the non-manifold spots stay where they are).  Passes over the (re-collected) edge list until a
// pass removes less than 1 % of the vertices.  Sequential host code, like the pass above; restated in oracle/meshing_oracle.py.
extern "C" int mvs_surface_simplify(mvs_surface *s, float min_angle_deg, float max_distance, mvs_simplify_report *report)
{
    if (!s || !(min_angle_deg >= 0.0f) || !(min_angle_deg < 60.0f) || !(max_distance >= 0.0f)) return MVS_EINVAL;
    const int nv = (int)(s->vertices.size() / 4), nf = (int)(s->faces.size() / 3);
    mvs_simplify_report r;
    std::memset(&r, 0, sizeof r);
    r.vertices_before = nv;
    r.facets_before = nf;
    try {
        Work w;
        w.p.resize((size_t)nv);
        for (int i = 0; i < nv; i++) w.p[(size_t)i] = {(double)s->vertices[4 * (size_t)i], (double)s->vertices[4 * (size_t)i + 1], (double)s->vertices[4 * (size_t)i + 2]};
        w.f = s->faces;
        w.inc.resize((size_t)nv);
        w.stamp.assign((size_t)nv, 0);
        for (int i = 0; i < nf; i++) {
            const int a = w.f[3 * i], b = w.f[3 * i + 1], c = w.f[3 * i + 2];
            if (a == b || b == c || a == c) {
                w.f[3 * i] = w.f[3 * i + 1] = w.f[3 * i + 2] = -1;
                continue;
            }
            w.inc[(size_t)a].push_back(i), w.inc[(size_t)b].push_back(i), w.inc[(size_t)c].push_back(i);
        }
        w.q_bound = -std::cos((double)min_angle_deg * 3.14159265358979323846 / 180.0);
        std::vector<double> err((size_t)nv, 0.0);
        // the input surface's normal at every vertex (unit; zero where it has none): facets in index order, so the sums are reproducible
        std::vector<V3> vnormal((size_t)nv, V3{0.0, 0.0, 0.0});
        for (int i = 0; i < nf; i++) {
            if (w.f[3 * i] < 0) continue;
            const V3 n = w.normal_of(w.f[3 * i], w.f[3 * i + 1], w.f[3 * i + 2]);
            for (int k = 0; k < 3; k++) {
                V3 &a = vnormal[(size_t)w.f[3 * i + k]];
                a = {a.x + n.x, a.y + n.y, a.z + n.z};
            }
        }
        for (V3 &a : vnormal) {
            const double l = std::sqrt(dot(a, a));
            a = l > 0.0 ? V3{a.x / l, a.y / l, a.z / l} : V3{0.0, 0.0, 0.0};
        }
        // how far the facets a collapse u -> v rewrites may lie from the input surface: max over them of ((L / 2) tan(theta / 2) + the largest err of
        // the corners); 1e300: a facet whose normal is more than ~78 degrees off a corner's kept normal
        auto deviation = [&](int u, int v) {
            int e[2];
            w.edge_facets(u, v, e);
            double worst = 0.0;
            for (int face : w.inc[(size_t)u]) {
                if (face == e[0] || face == e[1]) continue;
                int t[3];
                for (int k = 0; k < 3; k++) t[k] = w.f[3 * face + k] == u ? v : w.f[3 * face + k];
                const V3 n = w.normal_of(t[0], t[1], t[2]);
                const double ln = std::sqrt(dot(n, n));
                if (!(ln > 0.0)) return 1e300;
                double c = 1.0, emax = 0.0, l2 = 0.0;
                for (int k = 0; k < 3; k++) {
                    c = std::min(c, dot(n, vnormal[(size_t)t[k]]) / ln);
                    emax = std::max(emax, t[k] == v ? std::max(err[(size_t)v], err[(size_t)u]) : err[(size_t)t[k]]);
                    const V3 d = sub(w.p[(size_t)t[k]], w.p[(size_t)t[(k + 1) % 3]]);
                    l2 = std::max(l2, dot(d, d));
                }
                if (!(c > 0.2)) return 1e300;
                worst = std::max(worst, 0.5 * std::sqrt(l2) * std::sqrt(std::max(0.0, 1.0 - c * c)) / (1.0 + c) + emax);  // (L / 2) tan(theta / 2): the sagitta of a circular arc whose end normals are theta off the chord's
            }
            return worst;
        };
        // vertices that stay: on an edge with one facet (border) or more than two (non-manifold)
        std::vector<char> locked((size_t)nv, 0);
        for (int i = 0; i < nf; i++) {
            if (w.f[3 * i] < 0) continue;
            for (int k = 0; k < 3; k++) {
                const int u = w.f[3 * i + k], v = w.f[3 * i + (k + 1) % 3];
                int e[2];
                if (w.edge_facets(u, v, e) != 2) locked[(size_t)u] = locked[(size_t)v] = 1;
            }
        }
        struct Edge {
            double len2;
            int u, v;
        };
        std::vector<Edge> edges;
        std::vector<int> touched;
        int alive = 0;
        for (int i = 0; i < nv; i++) alive += !w.inc[(size_t)i].empty();
        for (int pass = 0; pass < 16; pass++) {
            edges.clear();
            for (int i = 0; i < nf; i++) {
                if (w.f[3 * i] < 0) continue;
                for (int k = 0; k < 3; k++) {
                    const int u = w.f[3 * i + k], v = w.f[3 * i + (k + 1) % 3];
                    if (u < v) {  // (a closed oriented surface has every edge once in each direction; on a border the u > v copy may be the only one: skipped, its ends are locked anyway)
                        const V3 d = sub(w.p[(size_t)u], w.p[(size_t)v]);
                        edges.push_back({dot(d, d), u, v});
                    }
                }
            }
            std::sort(edges.begin(), edges.end(), [](const Edge &a, const Edge &b) { return a.len2 != b.len2 ? a.len2 < b.len2 : (a.u != b.u ? a.u < b.u : a.v < b.v); });
            int removed = 0;
            for (const Edge &e : edges) {
                if (w.inc[(size_t)e.u].empty() || w.inc[(size_t)e.v].empty() || !w.connected(e.u, e.v)) continue;
                // both directions; the one that leaves the better worst angle among the facets it rewrites
                int best_u = -1, best_v = -1;
                double best_after = -2.0, best_near = 0.0;
                for (int dir = 0; dir < 2; dir++) {
                    const int u = dir ? e.v : e.u, v = dir ? e.u : e.v;
                    if (locked[(size_t)u]) continue;
                    w.guard = (double)max_distance - err[(size_t)u];
                    if (!(w.guard > 0.0)) continue;
                    double before = 0.0;
                    const double after = w.try_collapse(u, v, &before);
                    if (after >= w.q_bound && after > best_after && deviation(u, v) <= (double)max_distance) best_after = after, best_u = u, best_v = v, best_near = w.last_nearest;
                }
                if (best_u < 0) continue;
                touched.clear();
                w.do_collapse(best_u, best_v, touched);
                err[(size_t)best_v] = std::max(err[(size_t)best_v], err[(size_t)best_u] + best_near);
                removed++;
            }
            alive -= removed;
            if (removed * 100 < alive) break;
        }
        r.collapses = w.collapses;
        std::vector<int> renum((size_t)nv, -1);
        for (int i = 0; i < nf; i++)
            if (w.f[3 * i] >= 0)
                for (int k = 0; k < 3; k++) renum[(size_t)w.f[3 * i + k]] = 0;
        int kept_v = 0;
        std::vector<float> vout;
        double max_err = 0.0;
        for (int i = 0; i < nv; i++)
            if (renum[(size_t)i] == 0) {
                renum[(size_t)i] = kept_v++;
                max_err = std::max(max_err, err[(size_t)i]);
                for (int k = 0; k < 4; k++) vout.push_back(s->vertices[4 * (size_t)i + k]);
            }
        std::vector<int32_t> fout;
        for (int i = 0; i < nf; i++) {
            if (w.f[3 * i] < 0) continue;
            for (int k = 0; k < 3; k++) fout.push_back(renum[(size_t)w.f[3 * i + k]]);
        }
        r.vertices_after = kept_v;
        r.facets_after = (int)(fout.size() / 3);
        r.max_accumulated_distance = (float)max_err;
        s->vertices.swap(vout);
        s->faces.swap(fout);
    } catch (...) {
        return MVS_ENOMEM;
    }
    if (report) *report = r;
    return MVS_OK;
}
