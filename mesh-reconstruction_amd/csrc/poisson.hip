// poisson.hip -- poissonSurface (recon.hpp:37, cgal_poisson.cpp:47-136 / pcl.cpp:193-228 of the reference) on gfx950.
//
// The reference hands oriented points to CGAL's Poisson_reconstruction_function + make_surface_mesh (or, in pcl.cpp, to PCL's
// Poisson): third-party host code that is neither in this image nor vendored by the reference.  Both solve the same problem
// (Kazhdan, Bolitho, Hoppe: "Poisson surface reconstruction", SGP 2006): find the scalar field chi whose gradient best matches the
// field V of the samples' normals, laplace(chi) = div V, and extract the level set through the samples.  This is that method on a
// REGULAR grid sized for one MI355X (G^3 nodes, G <= 512: 0.5 GB per field at G = 512 out of 288 GB), which is the form the paper
// itself starts from (its section 3) before it introduces the octree to save memory:
//   0. the grid follows the reference's own yardstick: CGAL::compute_average_spacing(points, 6) (cgal_poisson.cpp:77 -- per sample the
//      mean distance to its 6 nearest neighbours, averaged over the samples), computed here on the device with a sorted cell grid and
//      an exact ring search, and the node spacing h is the largest 1.5 side / (2^k - 1), k = 5..9, with h <= 0.75 x that spacing:
//      measured on analytic surfaces the surface-nets vertices then stay within the reference's approximation bound of 0.375 x average
//      spacing (cgal_poisson.cpp:52, 99; tests/test_meshing_gpu.py); mvs_surface_spacing reports both numbers, and whether k = 9 was
//      too coarse to keep the ratio;
//   1. the samples' normals are splatted onto the grid nodes with trilinear weights -- 64-bit fixed-point atomics, so the field does
//      not depend on the order the atomics land in (bit-reproducible, and the CPU oracle gets the same integers) and a node cannot
//      overflow however many samples an outlier-stretched box packs into one cell (2^47 unit weights);
//   2. laplace(chi) = div V is solved in the Fourier domain (hipFFT: three real-to-complex transforms, chi^ = -i k.V^ / |k|^2 with a
//      Gaussian low-pass of `smooth` cells, one complex-to-real transform); periodic boundaries, kept away by padding the box;
//   3. the level: the (lower) median of chi (trilinear) over the samples, as CGAL shifts its implicit function to the median value at the input
//      points (Poisson_reconstruction_function::compute_implicit_function, called at cgal_poisson.cpp:72);
//   4a. only where the samples say something: cells within MVS_POISSON_SUPPORT_DEFAULT (8) average spacings of a node that collected
//      sample weight (a node mask, dilated by a max filter per axis).  Away from the samples chi is flat and hovers around the level;
//      on an open or noisy cloud the level set there is a closing sheet no sample supports plus numerical fuzz at full grid
//      resolution (measured on the config-5 cloud of the test-suite: 16 M vertices, 97 % of them tens of spacings from any sample).
//      mvs_poisson_surface_ex(support 0) returns the closed surface;
//   4. the level set is meshed by surface nets: one vertex per PATCH the level set cuts out of a grid cell, at the mean of the patch's
//      edge crossings (cell_patch_table: a cell that two sheets of the surface pass through gets two vertices -- with one vertex per cell
//      the sheets were welded there, edges with four facets that no facet criterion can repair; round 5); one quad = two triangles per
//      grid edge that changes sign, over the patches of its four cells that contain it; vertices and faces numbered in grid order by
//      exclusive scans (rocPRIM), faces oriented along +grad chi = the samples' normals (outward, like cgal_poisson.cpp:128-132).
// Output: vertices N x 4 homogeneous (w = 1), faces F x 3 int32, as Mesh (recon.hpp:19-24).  Not CGAL's triangulation: a different
// mesh of the same level set family (no Delaunay refinement; the reference's facet criteria -- angle, radius, distance, cgal_poisson.cpp:50-52 --
// are kept by the grid rule of step 0 and by mvs_surface_enforce_criteria, csrc/surface_criteria.cpp); DESIGN.md section 9 says so.
// Checked against oracle/poisson_oracle.py (numpy, float64) on the same inputs: identical splat integers, chi within 1e-4 of its
// range, surfaces within a fraction of a cell (tests/test_meshing_gpu.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <rocprim/rocprim.hpp>

#include "../../include/mvs.h"
#include "hooks.hpp"
#include "surface_internal.hpp"

namespace {

constexpr float SPLAT_SCALE = 65536.0f;  // fixed point of the splatted fields: 2^-16 per unit weight
constexpr float NORMAL_LIMIT = 1.0e4f;    // a normal component beyond this (or NaN) is no estimate: the sample votes for nothing (and n * w * 2^16 stays far inside int64 / exact in float)
constexpr int KNN = 6;                    // cgal_poisson.cpp:77: compute_average_spacing(points, 6)
typedef long long fix_t;

typedef SurfaceGrid Grid;  // csrc/surface_internal.hpp

__device__ __forceinline__ size_t node(const Grid &g, int i, int j, int k) { return ((size_t)k * g.G + j) * g.G + i; }

// samples: xyzw rows (w divides) + normal rows; nscale: the power of two that brings the normals' median length near 1 (exact)
__global__ void splat_kernel(Grid g, const float *__restrict__ pts, const float *__restrict__ nrm, int n, float nscale, fix_t *__restrict__ vx,
                             fix_t *__restrict__ vy, fix_t *__restrict__ vz, fix_t *__restrict__ wt)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float w = pts[4 * s + 3];
    const float gx = (pts[4 * s] / w - g.ox) / g.h, gy = (pts[4 * s + 1] / w - g.oy) / g.h, gz = (pts[4 * s + 2] / w - g.oz) / g.h;
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    if (i < 0 || j < 0 || k < 0 || i + 1 >= g.G || j + 1 >= g.G || k + 1 >= g.G) return;  // (the box is padded: cannot happen for the box's own samples)
    const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
    float nx = nrm[3 * s], ny = nrm[3 * s + 1], nz = nrm[3 * s + 2];
    if (!(fabsf(nx) <= NORMAL_LIMIT && fabsf(ny) <= NORMAL_LIMIT && fabsf(nz) <= NORMAL_LIMIT)) return;  // a sample without a usable normal (NaN: util.cpp:299's PCA can fail) votes for nothing
    nx *= nscale, ny *= nscale, nz *= nscale;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int di = c & 1, dj = (c >> 1) & 1, dk = c >> 2;
        const float wgt = (di ? tx : 1.0f - tx) * (dj ? ty : 1.0f - ty) * (dk ? tz : 1.0f - tz);
        const size_t q = node(g, i + di, j + dj, k + dk);
        // (two's complement: the unsigned 64-bit atomic adds signed values correctly)
        atomicAdd((unsigned long long *)&vx[q], (unsigned long long)(fix_t)rintf(nx * wgt * SPLAT_SCALE));
        atomicAdd((unsigned long long *)&vy[q], (unsigned long long)(fix_t)rintf(ny * wgt * SPLAT_SCALE));
        atomicAdd((unsigned long long *)&vz[q], (unsigned long long)(fix_t)rintf(nz * wgt * SPLAT_SCALE));
        atomicAdd((unsigned long long *)&wt[q], (unsigned long long)(fix_t)rintf(wgt * SPLAT_SCALE));
    }
}

// ---- average spacing (step 0): CGAL::compute_average_spacing(points, 6), cgal_poisson.cpp:77 -----------------------------------
// Per sample the mean distance to its KNN nearest OTHER samples (CGAL asks its k-d tree for k + 1 neighbours and skips the query point
// itself; coincident samples count, at distance 0), averaged over the samples.  On the device: the samples are sorted by the cell of
// a uniform grid (cell ~ two sample spacings of a surface sampling: a handful of samples per cell), a sample scans the block of
// (2R + 1)^3 cells around its own -- one binary search per cell row, the cells of a row are consecutive keys -- and stops as soon as
// its KNN-th distance is at most R cells: everything outside the block is farther than that.  R doubles otherwise (sparse corners,
// volumetric clouds).  Exact, not approximate -- up to a budget of KNN_BUDGET candidates per sample, which only a cloud with most of
// its samples inside one cell (densities many orders of magnitude apart) can exhaust: the distances found until then are used (an upper
// bound of the true ones), so that no input can turn the search into an n^2 scan that runs for minutes.  The host sums the per-sample
// means in sample order (deterministic).
constexpr int KNN_BUDGET = 1 << 17;
constexpr int KNN_ROW_COST = 16;  // what a probed cell row is charged, in candidates (two ~20-step binary searches)
struct CellGrid {
    float ox, oy, oz, inv, cell;
    int nx, ny, nz;
};

__device__ __forceinline__ void cell_of(const CellGrid &c, float x, float y, float z, int &i, int &j, int &k)
{
    i = min(max((int)floorf((x - c.ox) * c.inv), 0), c.nx - 1);
    j = min(max((int)floorf((y - c.oy) * c.inv), 0), c.ny - 1);
    k = min(max((int)floorf((z - c.oz) * c.inv), 0), c.nz - 1);
}

__global__ void cell_keys_kernel(CellGrid c, const float *__restrict__ pts, int n, unsigned *__restrict__ keys, int *__restrict__ ids)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float w = pts[4 * s + 3];
    int i, j, k;
    cell_of(c, pts[4 * s] / w, pts[4 * s + 1] / w, pts[4 * s + 2] / w, i, j, k);
    keys[s] = (unsigned)((k * c.ny + j) * c.nx + i);
    ids[s] = s;
}

__global__ void cell_points_kernel(const float *__restrict__ pts, const int *__restrict__ ids, int n, float4 *__restrict__ sorted)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int q = ids[s];
    const float w = pts[4 * q + 3];
    sorted[s] = make_float4(pts[4 * q] / w, pts[4 * q + 1] / w, pts[4 * q + 2] / w, 0.0f);
}

__device__ __forceinline__ int lower_bound_u32(const unsigned *__restrict__ a, int n, unsigned key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void knn_spacing_kernel(CellGrid c, const unsigned *__restrict__ keys, const float4 *__restrict__ sorted, const int *__restrict__ ids, int n,
                                   float *__restrict__ out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float4 p = sorted[s];
    int ci, cj, ck;
    cell_of(c, p.x, p.y, p.z, ci, cj, ck);
    float best[KNN];
    const int rmax = max(c.nx, max(c.ny, c.nz));
    int budget = KNN_BUDGET;
    for (int R = 1;; R *= 2) {
#pragma unroll
        for (int t = 0; t < KNN; t++) best[t] = 3.0e38f;
        const int i0 = max(ci - R, 0), i1 = min(ci + R, c.nx - 1);
        for (int k = max(ck - R, 0); k <= min(ck + R, c.nz - 1); k++)
            for (int j = max(cj - R, 0); j <= min(cj + R, c.ny - 1); j++) {
                const unsigned row = (unsigned)((k * c.ny + j) * c.nx);
                const int a = lower_bound_u32(keys, n, row + (unsigned)i0);
                int b = lower_bound_u32(keys, n, row + (unsigned)i1 + 1u);
                if (best[KNN - 1] == 0.0f || budget <= 0) b = a;  // nothing can come closer than coincident samples; or the budget is spent
                budget -= (b - a) + KNN_ROW_COST;  // a probed row costs its two binary searches even when it is empty (a far
                                                   // outlier -- homogeneous w near 0 -- walks (2R + 1)^2 empty rows per doubling of R)
                for (int t = a; t < b; t++) {
                    if (t == s) continue;
                    const float4 q = sorted[t];
                    const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
                    float d = dx * dx + dy * dy + dz * dz;
                    if (d < best[KNN - 1]) {  // insertion into the ascending list
#pragma unroll
                        for (int u = 0; u < KNN; u++) {
                            const float lo = fminf(best[u], d);
                            d = fmaxf(best[u], d);
                            best[u] = lo;
                        }
                    }
                }
            }
        const float reach = (float)R * c.cell;
        if (best[KNN - 1] <= reach * reach || R >= rmax || budget <= 0) break;
    }
    float sum = 0.0f;
    int m = 0;
#pragma unroll
    for (int t = 0; t < KNN; t++)
        if (best[t] < 3.0e38f) {
            sum += sqrtf(best[t]);
            m++;
        }
    // CGAL::compute_average_spacing(points, k) queries k + 1 neighbours -- the query point itself comes first, at distance 0 -- and divides the
    // sum of the distances by the number of points visited, k + 1 (compute_average_spacing.h; ADVICE r04: rounds 3-4 divided by k)
    out[ids[s]] = m ? sum / (float)(m + 1) : 0.0f;
}

__global__ void fixed_to_float_kernel(const fix_t *__restrict__ a, float *__restrict__ out, size_t n)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[q] = (float)a[q] * (1.0f / SPLAT_SCALE);
}

// chi^ = -i (k . V^) / |k|^2 * exp(-sigma^2 |k|^2 / 2) / G^3, k = 2 pi f / G per axis (f the signed frequency, 0 at the Nyquist
// frequency: the derivative of that mode is not defined on the grid); in place in sx
__global__ void spectral_solve_kernel(int G, float sigma, hipfftComplex *__restrict__ sx, const hipfftComplex *__restrict__ sy,
                                      const hipfftComplex *__restrict__ sz)
{
    const int H = G / 2 + 1;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t)G * G * H) return;
    const int a = (int)(q % H), b = (int)((q / H) % G), c = (int)(q / ((size_t)H * G));  // hipFFT's R2C layout: [z][y][x / 2 + 1]
    auto freq = [&](int f) { return f * 2 == G ? 0 : (f * 2 > G ? f - G : f); };
    const float two_pi_over_G = 6.28318530717958647692f / (float)G;
    const float kx = two_pi_over_G * (float)freq(a), ky = two_pi_over_G * (float)freq(b), kz = two_pi_over_G * (float)freq(c);
    const float k2 = kx * kx + ky * ky + kz * kz;
    hipfftComplex out = {0.0f, 0.0f};
    if (k2 > 0.0f) {
        const float re = kx * sx[q].x + ky * sy[q].x + kz * sz[q].x, im = kx * sx[q].y + ky * sy[q].y + kz * sz[q].y;
        const float s = expf(-0.5f * sigma * sigma * k2) / (k2 * (float)G * (float)G * (float)G);
        out.x = im * s;   // -i (re + i im) = im - i re
        out.y = -re * s;
    }
    sx[q] = out;
}

__device__ __forceinline__ float trilinear(const Grid &g, const float *__restrict__ f, float x, float y, float z)
{
    const float gx = (x - g.ox) / g.h, gy = (y - g.oy) / g.h, gz = (z - g.oz) / g.h;
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    int i = (int)fx, j = (int)fy, k = (int)fz;
    i = min(max(i, 0), g.G - 2), j = min(max(j, 0), g.G - 2), k = min(max(k, 0), g.G - 2);
    const float tx = gx - (float)i, ty = gy - (float)j, tz = gz - (float)k;
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int di = c & 1, dj = (c >> 1) & 1, dk = c >> 2;
        acc += (di ? tx : 1.0f - tx) * (dj ? ty : 1.0f - ty) * (dk ? tz : 1.0f - tz) * f[node(g, i + di, j + dj, k + dk)];
    }
    return acc;
}

__global__ void sample_kernel(Grid g, const float *__restrict__ chi, const float *__restrict__ pts, int n, float *__restrict__ out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float w = pts[4 * s + 3];
    out[s] = trilinear(g, chi, pts[4 * s] / w, pts[4 * s + 1] / w, pts[4 * s + 2] / w);
}

// cells: (G - 1)^3, cell (i, j, k) spans nodes i .. i + 1; "inside" = chi < iso (chi grows along the normals, which point out)
__device__ __forceinline__ size_t cell_id(int C, int i, int j, int k) { return ((size_t)k * C + j) * C + i; }

// ---- support of the samples (step 4a): a cell is meshed only where the samples say something about the surface ----
// node mask: 1 where a node within `R` nodes (Chebyshev distance, no wrap) collected any sample weight.  Seed, then one max-filter pass
// per axis (in -> out).  A cell belongs to the support when its low corner node does.
__global__ void support_seed_kernel(const fix_t *__restrict__ wt, unsigned char *__restrict__ m, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) m[i] = wt[i] > 0 ? 1 : 0;
}

__global__ void support_dilate_kernel(Grid g, const unsigned char *__restrict__ in, unsigned char *__restrict__ out, int R, int axis)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)g.G * g.G * g.G;
    if (q >= n) return;
    const int c[3] = {(int)(q % g.G), (int)((q / g.G) % g.G), (int)(q / ((size_t)g.G * g.G))};
    const size_t stride = axis == 0 ? 1 : (axis == 1 ? (size_t)g.G : (size_t)g.G * g.G);
    const int lo = max(0, c[axis] - R), hi = min(g.G - 1, c[axis] + R);
    const unsigned char *base = in + q - (size_t)c[axis] * stride;
    unsigned char v = 0;
    for (int t = lo; t <= hi && !v; t++) v = base[(size_t)t * stride];
    out[q] = v;
}

// The patches of a cell.  Entry m (bit c: corner c inside, corner bits 0 / 1 / 2 = x / y / z) packs, for each of the 12 cell edges in the
// order of `ea` / `eb` below, the number of the patch that contains its crossing (2 bits, edges that do not cross: 0) and in bits 24-26
// the number of patches (0..4).  Two crossings belong to one patch when a face of the cell joins them: a face with two crossings joins
// those two, a face with four (its two inside corners on a diagonal) joins the pair around EACH INSIDE corner -- the same decision seen
// from either cell of the face, so the patches of neighbouring cells meet edge for edge.  Patches are numbered by their lowest edge.
// (oracle/meshing_oracle.py: cell_components restates it.)
static void cell_patch_table(uint32_t table[256])
{
    static const int ea[12] = {0, 2, 4, 6, 0, 1, 4, 5, 0, 1, 2, 3}, eb[12] = {1, 3, 5, 7, 2, 3, 6, 7, 4, 5, 6, 7};
    for (int m = 0; m < 256; m++) {
        bool cross[12];
        int parent[12];
        for (int e = 0; e < 12; e++) {
            cross[e] = ((m >> ea[e]) & 1) != ((m >> eb[e]) & 1);
            parent[e] = e;
        }
        auto find = [&](int x) {
            while (parent[x] != x) x = parent[x];
            return x;
        };
        auto join = [&](int x, int y) {
            x = find(x), y = find(y);
            if (x != y) parent[std::max(x, y)] = std::min(x, y);
        };
        for (int axis = 0; axis < 3; axis++)
            for (int side = 0; side < 2; side++) {
                int es[4], ne = 0;
                for (int e = 0; e < 12; e++)
                    if (cross[e] && ((ea[e] >> axis) & 1) == side && ((eb[e] >> axis) & 1) == side) es[ne++] = e;
                if (ne == 2) join(es[0], es[1]);
                if (ne == 4)
                    for (int c = 0; c < 8; c++) {
                        if (((c >> axis) & 1) != side || !((m >> c) & 1)) continue;
                        int mine[2], nm = 0;
                        for (int i = 0; i < 4; i++)
                            if (ea[es[i]] == c || eb[es[i]] == c) mine[nm++] = es[i];
                        join(mine[0], mine[1]);
                    }
            }
        int roots[12], nr = 0;
        uint32_t word = 0;
        for (int e = 0; e < 12; e++) {
            if (!cross[e]) continue;
            const int r = find(e);  // (the root is the patch's lowest edge: first seen in this order)
            int id = 0;
            while (id < nr && roots[id] != r) id++;
            if (id == nr) roots[nr++] = r;
            word |= (uint32_t)id << (2 * e);
        }
        table[m] = word | ((uint32_t)nr << 24);
    }
}

// flag = the cell's number of vertices (patches; 0: not a mixed cell, or outside the samples' support); cases = its corner pattern
__global__ void cell_flags_kernel(Grid g, const float *__restrict__ chi, float iso, const unsigned char *__restrict__ support, const uint32_t *__restrict__ patches,
                                  int *__restrict__ flag, unsigned char *__restrict__ cases)
{
    const int C = g.G - 1;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t)C * C * C) return;
    const int i = (int)(q % C), j = (int)((q / C) % C), k = (int)(q / ((size_t)C * C));
    int m = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) m |= (chi[node(g, i + (c & 1), j + ((c >> 1) & 1), k + (c >> 2))] < iso ? 1 : 0) << c;
    cases[q] = (unsigned char)m;
    flag[q] = (m != 0 && m != 255 && (!support || support[node(g, i, j, k)])) ? (int)(patches[m] >> 24) : 0;
}

__global__ void cell_vertices_kernel(Grid g, const float *__restrict__ chi, float iso, const int *__restrict__ flag, const int *__restrict__ index,
                                     const uint32_t *__restrict__ patches, const unsigned char *__restrict__ cases, float *__restrict__ vertices)
{
    const int C = g.G - 1;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t)C * C * C || !flag[q]) return;
    const int i = (int)(q % C), j = (int)((q / C) % C), k = (int)(q / ((size_t)C * C));
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; c++) v[c] = chi[node(g, i + (c & 1), j + ((c >> 1) & 1), k + (c >> 2))] - iso;
    // the 12 edges of the cell as pairs of corner numbers (bit 0: x, bit 1: y, bit 2: z), in a fixed order
    const int ea[12] = {0, 2, 4, 6, 0, 1, 4, 5, 0, 1, 2, 3}, eb[12] = {1, 3, 5, 7, 2, 3, 6, 7, 4, 5, 6, 7};
    const uint32_t word = patches[cases[q]];
    const int np = flag[q];
    for (int patch = 0; patch < np; patch++) {
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        int m = 0;
#pragma unroll
        for (int e = 0; e < 12; e++) {
            const float a = v[ea[e]], b = v[eb[e]];
            if ((a < 0.0f) != (b < 0.0f) && (int)((word >> (2 * e)) & 3u) == patch) {
                const float t = a / (a - b);
                const int ca = ea[e], cb = eb[e];
                sx += (float)(ca & 1) + t * (float)((cb & 1) - (ca & 1));
                sy += (float)((ca >> 1) & 1) + t * (float)(((cb >> 1) & 1) - ((ca >> 1) & 1));
                sz += (float)(ca >> 2) + t * (float)((cb >> 2) - (ca >> 2));
                m++;
            }
        }
        const float inv = 1.0f / (float)m;
        float *out = vertices + 4 * ((size_t)index[q] + patch);
        out[0] = g.ox + g.h * ((float)i + sx * inv);
        out[1] = g.oy + g.h * ((float)j + sy * inv);
        out[2] = g.oz + g.h * ((float)k + sz * inv);
        out[3] = 1.0f;
    }
}

// grid edges: 3 per node (towards +x, +y, +z); an edge makes a quad when its ends lie on different sides and all four cells around
// it exist.  flag: 1 = inside -> outside along the axis, 2 = outside -> inside.
__device__ __forceinline__ int edge_state(const Grid &g, const float *__restrict__ chi, float iso, const unsigned char *__restrict__ support, int i, int j, int k, int axis)
{
    const int C = g.G - 1;
    const int i2 = i + (axis == 0), j2 = j + (axis == 1), k2 = k + (axis == 2);
    if (i2 > C || j2 > C || k2 > C) return 0;
    // the two axes across: cells at offsets -1 and 0 must exist
    const int u = axis == 0 ? j : (axis == 1 ? k : i), w = axis == 0 ? k : (axis == 1 ? i : j);
    if (u < 1 || u > C - 1 || w < 1 || w > C - 1) return 0;
    const bool a = chi[node(g, i, j, k)] < iso, b = chi[node(g, i2, j2, k2)] < iso;
    if (a == b) return 0;
    if (support) {  // ... and have a vertex: all four inside the samples' support (they are mixed cells: they contain this edge)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int du = (c == 0 || c == 3) ? -1 : 0, dw = c < 2 ? -1 : 0;
            int ci = i, cj = j, ck = k;
            if (axis == 0) cj += du, ck += dw;
            else if (axis == 1) ck += du, ci += dw;
            else ci += du, cj += dw;
            if (!support[node(g, ci, cj, ck)]) return 0;
        }
    }
    return a ? 1 : 2;
}

__global__ void edge_flags_kernel(Grid g, const float *__restrict__ chi, float iso, const unsigned char *__restrict__ support, int *__restrict__ flag)
{
    const size_t n = (size_t)g.G * g.G * g.G;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 3 * n) return;
    const int axis = (int)(q / n);
    const size_t r = q % n;
    const int i = (int)(r % g.G), j = (int)((r / g.G) % g.G), k = (int)(r / ((size_t)g.G * g.G));
    flag[q] = edge_state(g, chi, iso, support, i, j, k, axis) ? 1 : 0;
}

__global__ void edge_faces_kernel(Grid g, const float *__restrict__ chi, float iso, const unsigned char *__restrict__ support, const int *__restrict__ flag, const int *__restrict__ index,
                                  const int *__restrict__ cell_index, const uint32_t *__restrict__ patches, const unsigned char *__restrict__ cases, int *__restrict__ faces)
{
    const size_t n = (size_t)g.G * g.G * g.G;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 3 * n || !flag[q]) return;
    const int axis = (int)(q / n);
    const size_t r = q % n;
    const int i = (int)(r % g.G), j = (int)((r / g.G) % g.G), k = (int)(r / ((size_t)g.G * g.G));
    const int st = edge_state(g, chi, iso, support, i, j, k, axis);
    const int C = g.G - 1;
    // the four cells around the edge, counter-clockwise seen from the positive end of the axis: offsets (-1,-1), (0,-1), (0,0), (-1,0)
    // in the two axes (u, w) with axis = u x w: x: (y, z), y: (z, x), z: (x, y)
    const int du[4] = {-1, 0, 0, -1}, dw[4] = {-1, -1, 0, 0};
    int v[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int ci = i, cj = j, ck = k;
        int local;  // the grid edge inside that cell: x edges 0..3 = ly + 2 lz, y edges 4..7 = 4 + lx + 2 lz, z edges 8..11 = 8 + lx + 2 ly
        if (axis == 0) cj += du[c], ck += dw[c], local = -du[c] - 2 * dw[c];
        else if (axis == 1) ck += du[c], ci += dw[c], local = 4 - dw[c] - 2 * du[c];
        else ci += du[c], cj += dw[c], local = 8 - du[c] - 2 * dw[c];
        const size_t cell = cell_id(C, ci, cj, ck);
        v[c] = cell_index[cell] + (int)((patches[cases[cell]] >> (2 * local)) & 3u);  // the patch of that cell the edge's crossing belongs to
    }
    if (st == 2) {  // outside -> inside along the axis: the normal points down the axis
        const int t = v[1];
        v[1] = v[3];
        v[3] = t;
    }
    int *out = faces + 6 * (size_t)index[q];
    out[0] = v[0], out[1] = v[1], out[2] = v[2];
    out[3] = v[0], out[4] = v[2], out[5] = v[3];
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes)
    {
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return false;
        const bool poison = mvs::process_hooks().poison_alloc;  // (test hook, as in context.hip's ensure)
        return !poison || (hipMemset(p, 0xff, bytes ? bytes : 1) == hipSuccess);
    }
    template <class T> T *as() { return (T *)p; }
};

thread_local std::string g_poisson_error;
int fail(int code, const char *what)
{
    g_poisson_error = what;
    return code;
}

template <class K, class V>
bool sort_pairs(K *keys_in, K *keys_out, V *vals_in, V *vals_out, size_t n, hipStream_t st, void **tmp, size_t &tmp_bytes)
{
    size_t need = 0;
    if (rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, n, 0, sizeof(K) * 8, st) != hipSuccess) return false;
    if (need > tmp_bytes) {
        if (*tmp) (void)hipFree(*tmp);
        *tmp = nullptr;
        if (hipMalloc(tmp, need) != hipSuccess) return false;
        tmp_bytes = need;
    }
    return rocprim::radix_sort_pairs(*tmp, need, keys_in, keys_out, vals_in, vals_out, n, 0, sizeof(K) * 8, st) == hipSuccess;
}

bool scan(int *flags, int *offsets, size_t n, hipStream_t st, DevBuf &tmp, size_t &tmp_bytes)
{
    size_t need = 0;
    if (rocprim::exclusive_scan(nullptr, need, flags, offsets, 0, n, rocprim::plus<int>(), st) != hipSuccess) return false;
    if (need > tmp_bytes) {
        if (tmp.p) (void)hipFree(tmp.p);
        tmp.p = nullptr;
        if (!tmp.alloc(need)) return false;
        tmp_bytes = need;
    }
    return rocprim::exclusive_scan(tmp.p, need, flags, offsets, 0, n, rocprim::plus<int>(), st) == hipSuccess;
}

// hipFFT plans of the last grid size, kept for the process (rocFFT compiles its kernels for a size at the first
// plan of that size in a process -- about 2 s at 256^3 -- and building the two plans costs tens of milliseconds every time after that;
// Heuristic::tessellate calls this once per outer iteration with clouds of similar size, i.e. the same grid).  One size is kept (a plan
// holds work buffers); the mutex also serialises the transforms of concurrent callers on the shared plans.
struct PlanCache {
    std::mutex m;
    int G = 0, device = -1;  // (a plan belongs to the device it was made on)
    hipfftHandle fwd = 0, inv = 0;
};
PlanCache g_plans;

// the two plans of a G^3 grid on the current device into the cache (the caller holds g_plans.m); false: hipFFT refused
bool ensure_plans(int G, const char **why)
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) {
        *why = "hipGetDevice failed";
        return false;
    }
    if (g_plans.G == G && g_plans.device == device) return true;
    if (g_plans.G) {
        (void)hipfftDestroy(g_plans.fwd);
        (void)hipfftDestroy(g_plans.inv);
        g_plans.G = 0;
    }
    hipfftHandle fwd = 0, inv = 0;
    if (hipfftPlan3d(&fwd, G, G, G, HIPFFT_R2C) != HIPFFT_SUCCESS) {
        *why = "hipfftPlan3d (R2C) failed";
        return false;
    }
    if (hipfftPlan3d(&inv, G, G, G, HIPFFT_C2R) != HIPFFT_SUCCESS) {
        (void)hipfftDestroy(fwd);
        *why = "hipfftPlan3d (C2R) failed";
        return false;
    }
    g_plans.fwd = fwd, g_plans.inv = inv, g_plans.G = G, g_plans.device = device;
    return true;
}

}  // namespace

// test hook (not in mvs.h; tests/test_meshing_cpu.py): the patch table of the surface nets, as the kernels receive it (host code, no GPU)
extern "C" int mvs_test_cell_patch_table(uint32_t out[256])
{
    if (!out) return MVS_EINVAL;
    cell_patch_table(out);
    return MVS_OK;
}

extern "C" int mvs_poisson_warmup(int grid_log2)
{
    if (grid_log2 < 5 || grid_log2 > 9) {
        g_poisson_error = "mvs_poisson_warmup: grid_log2 out of range 5..9";
        return MVS_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_poisson_error = "mvs_poisson_warmup: no HIP device available";
        return MVS_EHIP;
    }
    std::lock_guard<std::mutex> lock(g_plans.m);
    const char *why = "";
    if (!ensure_plans(1 << grid_log2, &why)) {
        g_poisson_error = std::string("mvs_poisson_warmup: ") + why;
        return MVS_EHIP;
    }
    return MVS_OK;
}

extern "C" const char *mvs_surface_last_error(void) { return g_poisson_error.c_str(); }

extern "C" int mvs_poisson_surface(const float *points, const float *normals, int n, int grid_log2, float smooth_cells, int keep_fields, mvs_surface **out)
{
    return mvs_poisson_surface_ex(points, normals, n, grid_log2, smooth_cells, MVS_POISSON_SUPPORT_DEFAULT, keep_fields, out);
}

extern "C" int mvs_poisson_surface_ex(const float *points, const float *normals, int n, int grid_log2, float smooth_cells, float support_spacings, int keep_fields,
                                      mvs_surface **out)
{
    if (!points || !normals || n < 1 || !out || grid_log2 < 0 || grid_log2 > 9 || !(smooth_cells >= 0.0f) || !(support_spacings >= 0.0f) || support_spacings > 1.0e6f)
        return fail(MVS_EINVAL, "mvs_poisson_surface: bad argument");
    *out = nullptr;
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) return fail(MVS_EHIP, "mvs_poisson_surface: no HIP device (this library has no CPU path)");
    // the box: the samples' bounding cube, padded by a quarter of its side on every face (periodic solve: keep the wrap-around away)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int s = 0; s < n; s++) {
        const float w = points[4 * s + 3];
        for (int c = 0; c < 3; c++) {
            const double v = (double)(points[4 * s + c] / w);
            if (!(v == v) || std::fabs(v) > 1e30) return fail(MVS_EINVAL, "mvs_poisson_surface: a point is not finite");
            lo[c] = std::min(lo[c], v), hi[c] = std::max(hi[c], v);
        }
    }
    double side = 0.0;
    for (int c = 0; c < 3; c++) side = std::max(side, hi[c] - lo[c]);
    if (!(side > 0.0)) return fail(MVS_EINVAL, "mvs_poisson_surface: the points have no extent");
    const double box = 1.5 * side;
    // ---- step 0: the samples' average spacing (device), then the grid ----
    // a stream of its own (not the legacy default stream, which synchronises with every blocking stream of the process)
    struct OwnStream {
        hipStream_t s = nullptr;
        ~OwnStream() { if (s) (void)hipStreamDestroy(s); }
    } own;
    if (hipStreamCreateWithFlags(&own.s, hipStreamNonBlocking) != hipSuccess) return fail(MVS_EHIP, "mvs_poisson_surface: hipStreamCreate failed");
    hipStream_t st = own.s;
    DevBuf d_pts, d_nrm;
    if (!d_pts.alloc((size_t)n * 16) || !d_nrm.alloc((size_t)n * 12)) return fail(MVS_ENOMEM, "mvs_poisson_surface: device allocation failed");
    if (hipMemcpy(d_pts.p, points, (size_t)n * 16, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_nrm.p, normals, (size_t)n * 12, hipMemcpyHostToDevice) != hipSuccess)
        return fail(MVS_EHIP, "mvs_poisson_surface: upload failed");
    double spacing = 0.0;
    {
        // The cell grid spans a ROBUST extent of the samples (2nd .. 98th percentile per axis, plus a margin): a few outliers must not
        // blow the cells up until the real samples share one of them (the search stays exact either way: samples outside the grid are
        // binned into its border cells, and a border cell is only skipped when it is farther away than the search radius).
        CellGrid cg;
        double rlo[3], rhi[3], rside = 0.0;
        try {
            std::vector<float> axis((size_t)n);
            for (int c = 0; c < 3; c++) {
                for (int s = 0; s < n; s++) axis[(size_t)s] = points[4 * s + c] / points[4 * s + 3];
                const size_t a = (size_t)(0.02 * (double)(n - 1)), b = (size_t)(0.98 * (double)(n - 1));
                std::nth_element(axis.begin(), axis.begin() + (std::ptrdiff_t)a, axis.end());
                rlo[c] = (double)axis[a];
                std::nth_element(axis.begin(), axis.begin() + (std::ptrdiff_t)b, axis.end());
                rhi[c] = (double)axis[b];
                rside = std::max(rside, rhi[c] - rlo[c]);
            }
        } catch (...) {
            return fail(MVS_ENOMEM, "mvs_poisson_surface: host allocation failed");
        }
        if (!(rside > 0.0)) {  // degenerate percentiles (most samples coincide): the full box
            rside = side;
            for (int c = 0; c < 3; c++) rlo[c] = lo[c], rhi[c] = hi[c];
        }
        const double cell = std::max(2.0 * rside / std::sqrt((double)n), rside / 1000.0);  // ~ two spacings of a surface sampling; at most ~1000 cells per axis
        cg.cell = (float)cell;
        cg.inv = (float)(1.0 / cell);
        cg.ox = (float)(rlo[0] - 2.0 * cell), cg.oy = (float)(rlo[1] - 2.0 * cell), cg.oz = (float)(rlo[2] - 2.0 * cell);
        cg.nx = std::max(1, std::min(1024, (int)std::floor((rhi[0] - rlo[0]) / cell) + 5));
        cg.ny = std::max(1, std::min(1024, (int)std::floor((rhi[1] - rlo[1]) / cell) + 5));
        cg.nz = std::max(1, std::min(1024, (int)std::floor((rhi[2] - rlo[2]) / cell) + 5));
        DevBuf k0, k1, i0, i1, sorted, per;
        void *tmp = nullptr;
        size_t tmp_bytes = 0;
        bool ok = k0.alloc((size_t)n * 4) && k1.alloc((size_t)n * 4) && i0.alloc((size_t)n * 4) && i1.alloc((size_t)n * 4) && sorted.alloc((size_t)n * 16) && per.alloc((size_t)n * 4);
        std::vector<float> host;
        try {
            host.resize((size_t)n);
        } catch (...) {
            ok = false;
        }
        if (ok) {
            cell_keys_kernel<<<(n + 255) / 256, 256, 0, st>>>(cg, d_pts.as<float>(), n, k0.as<unsigned>(), i0.as<int>());
            ok = sort_pairs(k0.as<unsigned>(), k1.as<unsigned>(), i0.as<int>(), i1.as<int>(), (size_t)n, st, &tmp, tmp_bytes);
        }
        if (ok) {
            cell_points_kernel<<<(n + 255) / 256, 256, 0, st>>>(d_pts.as<float>(), i1.as<int>(), n, sorted.as<float4>());
            knn_spacing_kernel<<<(n + 127) / 128, 128, 0, st>>>(cg, k1.as<unsigned>(), sorted.as<float4>(), i1.as<int>(), n, per.as<float>());
            ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(host.data(), per.p, (size_t)n * 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        }
        if (tmp) (void)hipFree(tmp);
        if (!ok) return fail(MVS_EHIP, "mvs_poisson_surface: average spacing failed");
        for (int s = 0; s < n; s++) spacing += (double)host[s];
        spacing /= (double)n;
    }
    // ---- the normals' scale.  Their LENGTHS are confidences (triangulatePixels multiplies by a pdf, util.cpp:322-327: lengths of 1e-6 .. 1e-4
    // on real frames) and only their ratios matter to the level set; the splat is fixed point (2^-16), so the normals are first multiplied
    // by the power of two that brings the median of max(|nx|, |ny|, |nz|) over the usable, non-zero normals into [0.5, 1) -- exact, and
    // 1 for unit normals -- kept low enough that the largest usable component stays within NORMAL_LIMIT.
    int nscale_log2 = 0;
    try {
        std::vector<float> big;
        big.reserve((size_t)n);
        float largest = 0.0f;
        for (int s = 0; s < n; s++) {
            const float a = std::fabs(normals[3 * s]), b = std::fabs(normals[3 * s + 1]), c = std::fabs(normals[3 * s + 2]);
            if (!(a <= NORMAL_LIMIT && b <= NORMAL_LIMIT && c <= NORMAL_LIMIT)) continue;
            const float m = std::max(a, std::max(b, c));
            if (m > 0.0f) big.push_back(m), largest = std::max(largest, m);
        }
        if (!big.empty()) {
            const size_t mid = (big.size() - 1) / 2;  // the lower median
            std::nth_element(big.begin(), big.begin() + (std::ptrdiff_t)mid, big.end());
            int e = 0;
            (void)std::frexp(big[mid], &e);  // median = m 2^e, m in [0.5, 1)
            nscale_log2 = std::max(-100, std::min(100, -e));
            while (nscale_log2 > -100 && std::ldexp((double)largest, nscale_log2) > (double)NORMAL_LIMIT) nscale_log2--;
        }
    } catch (...) {
        return fail(MVS_ENOMEM, "mvs_poisson_surface: host allocation failed");
    }
    int lg = grid_log2, ratio_kept = 1;
    if (lg == 0) {  // the coarsest grid whose nodes are at most 0.75 average spacings apart (header: step 0); 512^3 at the most
        lg = 5;
        while (lg < 9 && box / (double)((1 << lg) - 1) > 0.75 * spacing) lg++;
        ratio_kept = box / (double)((1 << lg) - 1) <= 0.75 * spacing ? 1 : 0;
    } else {
        ratio_kept = box / (double)((1 << lg) - 1) <= 0.75 * spacing ? 1 : 0;
    }
    if (lg < 4) return fail(MVS_EINVAL, "mvs_poisson_surface: the grid needs at least 16 nodes per axis");
    Grid g;
    g.G = 1 << lg;
    g.h = (float)(box / (double)(g.G - 1));
    g.ox = (float)(0.5 * (lo[0] + hi[0]) - 0.5 * box);
    g.oy = (float)(0.5 * (lo[1] + hi[1]) - 0.5 * box);
    g.oz = (float)(0.5 * (lo[2] + hi[2]) - 0.5 * box);
    const size_t N3 = (size_t)g.G * g.G * g.G, S3 = (size_t)g.G * g.G * (g.G / 2 + 1);
    const int C = g.G - 1;
    const size_t C3 = (size_t)C * C * C;

    mvs_surface *res = new (std::nothrow) mvs_surface;
    if (!res) return fail(MVS_ENOMEM, "mvs_poisson_surface: host allocation failed");
    res->grid = g;
    res->normal_scale_log2 = nscale_log2;
    res->spacing = (float)spacing;
    res->ratio_kept = ratio_kept;
    DevBuf d_fix, d_real, d_spec, d_flag, d_index, d_cell_index, d_tmp, d_samples, d_vertices, d_faces, d_mask, d_patches, d_cases;
    size_t tmp_bytes = 0;
    int rc = MVS_OK;
    const char *msg = "";
#define PS_TRY(cond, code, text) do { if (!(cond)) { rc = (code); msg = (text); goto done; } } while (0)
    try {
        PS_TRY(d_fix.alloc(4 * N3 * sizeof(fix_t)) && d_real.alloc(3 * N3 * 4) && d_spec.alloc(3 * S3 * 8) &&
                   d_flag.alloc(3 * N3 * 4) && d_index.alloc((3 * N3 + 1) * 4) && d_cell_index.alloc((C3 + 1) * 4) && d_samples.alloc((size_t)n * 4),
               MVS_ENOMEM, "mvs_poisson_surface: device allocation failed");
        PS_TRY(hipMemsetAsync(d_fix.p, 0, 4 * N3 * sizeof(fix_t), st) == hipSuccess, MVS_EHIP, "mvs_poisson_surface: clearing the grid failed");
        fix_t *vx = d_fix.as<fix_t>(), *vy = vx + N3, *vz = vy + N3, *wt = vz + N3;
        splat_kernel<<<(n + 255) / 256, 256, 0, st>>>(g, d_pts.as<float>(), d_nrm.as<float>(), n, std::ldexp(1.0f, nscale_log2), vx, vy, vz, wt);
        fixed_to_float_kernel<<<(unsigned)((3 * N3 + 255) / 256), 256, 0, st>>>(vx, d_real.as<float>(), 3 * N3);
        PS_TRY(hipGetLastError() == hipSuccess, MVS_EHIP, "mvs_poisson_surface: splat launch failed");
        hipfftComplex *spec = d_spec.as<hipfftComplex>();
        float *chi = d_real.as<float>();  // (the transform may overwrite its input: spec[0] is not used again)
        {
            std::lock_guard<std::mutex> lock(g_plans.m);  // the plans of the last grid size are kept for the process (see PlanCache)
            const char *why = "";
            if (!ensure_plans(g.G, &why)) PS_TRY(false, MVS_EHIP, why);
            PS_TRY(hipfftSetStream(g_plans.fwd, st) == HIPFFT_SUCCESS && hipfftSetStream(g_plans.inv, st) == HIPFFT_SUCCESS, MVS_EHIP, "mvs_poisson_surface: hipfftSetStream failed");
            for (int c = 0; c < 3; c++)
                PS_TRY(hipfftExecR2C(g_plans.fwd, d_real.as<float>() + c * N3, spec + c * S3) == HIPFFT_SUCCESS, MVS_EHIP, "mvs_poisson_surface: forward FFT failed");
            spectral_solve_kernel<<<(unsigned)((S3 + 255) / 256), 256, 0, st>>>(g.G, smooth_cells, spec, spec + S3, spec + 2 * S3);
            PS_TRY(hipfftExecC2R(g_plans.inv, spec, chi) == HIPFFT_SUCCESS, MVS_EHIP, "mvs_poisson_surface: inverse FFT failed");
            PS_TRY(hipStreamSynchronize(st) == hipSuccess, MVS_EHIP, "mvs_poisson_surface: the transforms failed");  // before another caller may use the plans
        }
        sample_kernel<<<(n + 255) / 256, 256, 0, st>>>(g, chi, d_pts.as<float>(), n, d_samples.as<float>());
        std::vector<float> samples((size_t)n);
        PS_TRY(hipMemcpyAsync(samples.data(), d_samples.p, (size_t)n * 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess, MVS_EHIP,
               "mvs_poisson_surface: sampling failed");
        // the level: the MEDIAN of chi over the samples, as CGAL's Poisson_reconstruction_function shifts its function ("f() = 0 on the input
        // points": median_value_at_input_vertices) -- the lower median; with confidences that span two decades the mean follows the few strong samples
        std::nth_element(samples.begin(), samples.begin() + (std::ptrdiff_t)((n - 1) / 2), samples.end());
        const float iso = samples[(size_t)((n - 1) / 2)];
        res->iso = iso;
        // the samples' support: nodes within support_spacings average spacings of a node that collected weight (0: everywhere)
        const unsigned char *support = nullptr;
        if (support_spacings > 0.0f && spacing > 0.0) {
            const int R = (int)std::min((double)g.G, std::ceil((double)support_spacings * spacing / (double)g.h));
            res->support_cells = R;
            PS_TRY(d_mask.alloc(2 * N3), MVS_ENOMEM, "mvs_poisson_surface: device allocation failed");
            unsigned char *m0 = d_mask.as<unsigned char>(), *m1 = m0 + N3;
            support_seed_kernel<<<(unsigned)((N3 + 255) / 256), 256, 0, st>>>(wt, m0, N3);
            support_dilate_kernel<<<(unsigned)((N3 + 255) / 256), 256, 0, st>>>(g, m0, m1, R, 0);
            support_dilate_kernel<<<(unsigned)((N3 + 255) / 256), 256, 0, st>>>(g, m1, m0, R, 1);
            support_dilate_kernel<<<(unsigned)((N3 + 255) / 256), 256, 0, st>>>(g, m0, m1, R, 2);
            PS_TRY(hipGetLastError() == hipSuccess, MVS_EHIP, "mvs_poisson_surface: support mask launch failed");
            support = m1;
        }
        // vertices
        int *flag = d_flag.as<int>(), *cell_index = d_cell_index.as<int>();
        uint32_t patch_table[256];
        cell_patch_table(patch_table);
        PS_TRY(d_patches.alloc(sizeof(patch_table)) && d_cases.alloc(C3), MVS_ENOMEM, "mvs_poisson_surface: device allocation failed");
        PS_TRY(hipMemcpyAsync(d_patches.p, patch_table, sizeof(patch_table), hipMemcpyHostToDevice, st) == hipSuccess, MVS_EHIP, "mvs_poisson_surface: upload failed");
        const uint32_t *patches = d_patches.as<uint32_t>();
        const unsigned char *cases = d_cases.as<unsigned char>();
        cell_flags_kernel<<<(unsigned)((C3 + 255) / 256), 256, 0, st>>>(g, chi, iso, support, patches, flag, d_cases.as<unsigned char>());
        PS_TRY(hipMemsetAsync(flag + C3, 0, 4, st) == hipSuccess && scan(flag, cell_index, C3 + 1, st, d_tmp, tmp_bytes), MVS_EHIP, "mvs_poisson_surface: scan failed");
        int nv = 0;
        PS_TRY(hipMemcpyAsync(&nv, cell_index + C3, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess, MVS_EHIP, "mvs_poisson_surface: count failed");
        PS_TRY(d_vertices.alloc((size_t)nv * 16), MVS_ENOMEM, "mvs_poisson_surface: device allocation failed");
        if (nv > 0) cell_vertices_kernel<<<(unsigned)((C3 + 255) / 256), 256, 0, st>>>(g, chi, iso, flag, cell_index, patches, cases, d_vertices.as<float>());
        res->vertices.resize((size_t)nv * 4);
        if (nv > 0) PS_TRY(hipMemcpyAsync(res->vertices.data(), d_vertices.p, (size_t)nv * 16, hipMemcpyDeviceToHost, st) == hipSuccess, MVS_EHIP, "mvs_poisson_surface: download failed");
        // faces (the cell flags are overwritten by the edge flags: the vertex kernel above is ordered before on the stream)
        int *index = d_index.as<int>();
        edge_flags_kernel<<<(unsigned)((3 * N3 + 255) / 256), 256, 0, st>>>(g, chi, iso, support, flag);
        {
            // flag has 3 N3 entries; the scan needs one more (the total): d_flag was sized 3 N3, so scan into index[0 .. 3 N3] with the
            // total computed from the last offset + last flag
            PS_TRY(scan(flag, index, 3 * N3, st, d_tmp, tmp_bytes), MVS_EHIP, "mvs_poisson_surface: scan failed");
        }
        int last_off = 0, last_flag = 0;
        PS_TRY(hipMemcpyAsync(&last_off, index + 3 * N3 - 1, 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                   hipMemcpyAsync(&last_flag, flag + 3 * N3 - 1, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess,
               MVS_EHIP, "mvs_poisson_surface: count failed");
        const int nq = last_off + last_flag;
        PS_TRY(d_faces.alloc((size_t)nq * 24), MVS_ENOMEM, "mvs_poisson_surface: device allocation failed");
        if (nq > 0) edge_faces_kernel<<<(unsigned)((3 * N3 + 255) / 256), 256, 0, st>>>(g, chi, iso, support, flag, index, cell_index, patches, cases, d_faces.as<int>());
        res->faces.resize((size_t)nq * 6);
        if (nq > 0) PS_TRY(hipMemcpyAsync(res->faces.data(), d_faces.p, (size_t)nq * 24, hipMemcpyDeviceToHost, st) == hipSuccess, MVS_EHIP, "mvs_poisson_surface: download failed");
        if (keep_fields) {
            res->chi.resize(N3);
            res->splat.resize(4 * N3);
            PS_TRY(hipMemcpyAsync(res->chi.data(), chi, N3 * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                       hipMemcpyAsync(res->splat.data(), d_fix.p, 4 * N3 * sizeof(fix_t), hipMemcpyDeviceToHost, st) == hipSuccess,
                   MVS_EHIP, "mvs_poisson_surface: download failed");
        }
        PS_TRY(hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess, MVS_EHIP, "mvs_poisson_surface: a kernel failed");
    } catch (...) {  // a host vector could not grow: no exception crosses the C boundary
        rc = MVS_ENOMEM;
        msg = "mvs_poisson_surface: host allocation failed";
    }
done:
#undef PS_TRY
    if (rc != MVS_OK) {
        delete res;
        return fail(rc, msg);
    }
    *out = res;
    return MVS_OK;
}

extern "C" int mvs_surface_counts(const mvs_surface *s, int *vertices, int *faces)
{
    if (!s) return MVS_EINVAL;
    if (vertices) *vertices = (int)(s->vertices.size() / 4);
    if (faces) *faces = (int)(s->faces.size() / 3);
    return MVS_OK;
}

extern "C" int mvs_surface_fetch(const mvs_surface *s, float *vertices, int32_t *faces)
{
    if (!s) return MVS_EINVAL;
    if (vertices && !s->vertices.empty()) std::memcpy(vertices, s->vertices.data(), s->vertices.size() * 4);
    if (faces && !s->faces.empty()) std::memcpy(faces, s->faces.data(), s->faces.size() * 4);
    return MVS_OK;
}

extern "C" int mvs_surface_spacing(const mvs_surface *s, float *average_spacing, float *node_spacing, int *ratio_kept)
{
    if (!s) return MVS_EINVAL;
    if (average_spacing) *average_spacing = s->spacing;
    if (node_spacing) *node_spacing = s->grid.h;
    if (ratio_kept) *ratio_kept = s->ratio_kept;
    return MVS_OK;
}

extern "C" int mvs_surface_support(const mvs_surface *s, int *support_nodes)
{
    if (!s) return MVS_EINVAL;
    if (support_nodes) *support_nodes = s->support_cells;
    return MVS_OK;
}

extern "C" int mvs_surface_normal_scale(const mvs_surface *s, int *scale_log2)
{
    if (!s) return MVS_EINVAL;
    if (scale_log2) *scale_log2 = s->normal_scale_log2;
    return MVS_OK;
}

extern "C" int mvs_surface_grid(const mvs_surface *s, int *nodes_per_axis, float *origin3, float *spacing, float *level, float *chi, int64_t *splat)
{
    if (!s) return MVS_EINVAL;
    if (nodes_per_axis) *nodes_per_axis = s->grid.G;
    if (origin3) origin3[0] = s->grid.ox, origin3[1] = s->grid.oy, origin3[2] = s->grid.oz;
    if (spacing) *spacing = s->grid.h;
    if (level) *level = s->iso;
    if (chi) {
        if (s->chi.empty()) return MVS_ESTATE;
        std::memcpy(chi, s->chi.data(), s->chi.size() * 4);
    }
    if (splat) {
        if (s->splat.empty()) return MVS_ESTATE;
        std::memcpy(splat, s->splat.data(), s->splat.size() * sizeof(int64_t));
    }
    return MVS_OK;
}

extern "C" void mvs_surface_free(mvs_surface *s) { delete s; }
