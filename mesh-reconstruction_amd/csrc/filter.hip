// filter.hip -- Heuristic::filterPoints (heuristic.cpp:55-176) on gfx950: the outlier / redundancy filter the reference
// runs on the 10^5..10^6-point cloud after every iteration (recon.cpp:125).
//
// Reference structure -> here (everything on the device; the host only reads an 8-byte convergence value per power
// iteration, a round counter per batch of greedy rounds, and the final keep flags):
//   FLANN KD-tree build + one radiusSearch per point (heuristic.cpp:74-92, single thread, randomised, approximate)
//       -> hash grid with cell = sqrt(radius): hash_build, nb_count, nb_fill (exact neighbourhood, 27 cells per point);
//          every point gets its lower list (j < i) and its upper list (k > i), both ascending by index so the result does
//          not depend on atomics' arrival order -- ordered per thread for short lists, by global radix sorts with key
//          (owner, index) for long ones (list_keys)
//   power iteration over the symmetric neighbour weights (103-136), scatter `score[j] += ...` on one thread
//       -> gather form: a point adds its lower and upper lists in exactly the order the sequential scatter would have
//          produced (density_score); the two global sums are fixed-shape reductions finished on the device
//          (chunk_sums, finish_normalizer / finish_change); iteration k + 1 is queued before the host sees iteration k's value
//   greedy pass by descending density (139-163)
//       -> the order is a radix sort of (~density bits, index) keys; the pass itself only looks serial: a point depends on the
//          kept points earlier in the order whose lower list contains it, so it is decided in dependency rounds
//          (greedy_round / greedy_round_sorted), exactly the sequential result; a bounded host walk finishes pathological chains
// Quirks kept: squared distances compared with `radius` (81-89), only lower-index neighbours are penalised (152-154).
#include "mvs_internal.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <numeric>

namespace mvs {

struct Nb {
    int idx;
    float w;
};

__device__ __forceinline__ unsigned cell_hash(int cx, int cy, int cz, unsigned mask)
{
    return ((unsigned)cx * 73856093u ^ (unsigned)cy * 19349663u ^ (unsigned)cz * 83492791u) & mask;
}

__global__ __launch_bounds__(256) void dehomog_cells(const float *__restrict__ p4, int N, float inv_cell, float *__restrict__ p3,
                                                     int *__restrict__ cell3)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float v = p4[4 * i + c] / p4[4 * i + 3];  // dehomogenize, util.cpp:16-29
        p3[3 * i + c] = v;
        const float f = floorf(v * inv_cell);
        cell3[3 * i + c] = (int)fminf(fmaxf(f, -1.0e9f), 1.0e9f);
    }
}

__global__ __launch_bounds__(256) void hash_build(const int *__restrict__ cell3, int N, unsigned mask, int *__restrict__ head,
                                                  int *__restrict__ next)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const unsigned h = cell_hash(cell3[3 * i], cell3[3 * i + 1], cell3[3 * i + 2], mask);
    next[i] = atomicExch(&head[h], i);
}

// visit every j < i with |p_i - p_j|^2 <= radius; F(j, d2)
template <class F>
__device__ __forceinline__ void for_lower_neighbours(int i, const float *__restrict__ p3, const int *__restrict__ cell3,
                                                     const int *__restrict__ head, const int *__restrict__ next, unsigned mask,
                                                     float radius, F f)
{
    const float x = p3[3 * i], y = p3[3 * i + 1], z = p3[3 * i + 2];
    const int cx = cell3[3 * i], cy = cell3[3 * i + 1], cz = cell3[3 * i + 2];
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int nx = cx + dx, ny = cy + dy, nz = cz + dz;
                for (int j = head[cell_hash(nx, ny, nz, mask)]; j >= 0; j = next[j]) {
                    // buckets mix cells: accept only points of THIS cell, so no pair is seen twice
                    if (j >= i || cell3[3 * j] != nx || cell3[3 * j + 1] != ny || cell3[3 * j + 2] != nz) continue;
                    const float ddx = x - p3[3 * j], ddy = y - p3[3 * j + 1], ddz = z - p3[3 * j + 2];
                    const float d2 = ddx * ddx + ddy * ddy + ddz * ddz;
                    if (d2 <= radius) f(j, d2);
                }
            }
}

// 1 into *flag when a prefix-sum array of N + 1 non-negative counts decreases somewhere (its 32-bit sum wrapped)
__global__ void offsets_wrap_check(const int *__restrict__ off_lo, const int *__restrict__ off_up, int N, int *__restrict__ flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && (off_lo[i + 1] < off_lo[i] || off_up[i + 1] < off_up[i] || off_lo[i] < 0 || off_up[i] < 0)) *flag = 1;
}

__global__ __launch_bounds__(256) void nb_count(const float *__restrict__ p3, const int *__restrict__ cell3,
                                                const int *__restrict__ head, const int *__restrict__ next, unsigned mask, int N,
                                                float radius, int *__restrict__ cnt_lo, int *__restrict__ cnt_up)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int c = 0;
    for_lower_neighbours(i, p3, cell3, head, next, mask, radius, [&](int j, float) {
        c++;
        atomicAdd(&cnt_up[j], 1);
    });
    cnt_lo[i] = c;
}

// ORDERED: keep each lower list ascending by insertion while filling (short lists); otherwise append in traversal order
// and leave the ordering to the global sort (long lists: the insertion is O(L^2) global-memory moves per point)
template <bool ORDERED>
__global__ __launch_bounds__(256) void nb_fill(const float *__restrict__ p3, const int *__restrict__ cell3,
                                               const int *__restrict__ head, const int *__restrict__ next, unsigned mask, int N,
                                               float radius, const int *__restrict__ off_lo, const int *__restrict__ off_up,
                                               int *__restrict__ fill_up, Nb *__restrict__ lo, Nb *__restrict__ up)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Nb *mine = lo + off_lo[i];
    int c = 0;
    for_lower_neighbours(i, p3, cell3, head, next, mask, radius, [&](int j, float d2) {
        const float w = (float)(1. - d2 / radius);  // densityFn, heuristic.cpp:49-52
        // insertion into the ascending lower list
        int k = c++;
        while (ORDERED && k > 0 && mine[k - 1].idx > j) {
            mine[k] = mine[k - 1];
            k--;
        }
        mine[k].idx = j;
        mine[k].w = w;
        const int pos = atomicAdd(&fill_up[j], 1);
        up[off_up[j] + pos].idx = i;
        up[off_up[j] + pos].w = w;
    });
}

// upper lists arrive in atomic order: sort each by index
__global__ __launch_bounds__(256) void sort_upper(const int *__restrict__ off_up, int N, Nb *__restrict__ up)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Nb *a = up + off_up[i];
    const int n = off_up[i + 1] - off_up[i];
    for (int k = 1; k < n; k++) {
        const Nb t = a[k];
        int m = k;
        while (m > 0 && a[m - 1].idx > t.idx) {
            a[m] = a[m - 1];
            m--;
        }
        a[m] = t;
    }
}

// one power-iteration round, gather form of heuristic.cpp:107-120
__global__ __launch_bounds__(256) void density_score(const float *__restrict__ density, const int *__restrict__ off_lo,
                                                     const Nb *__restrict__ lo, const int *__restrict__ off_up,
                                                     const Nb *__restrict__ up, int N, float *__restrict__ score,
                                                     double *__restrict__ pair_sum)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float di = density[i];
    float densityTemp = 0.f;
    double ps = 0.;
    for (int k = off_lo[i]; k < off_lo[i + 1]; k++) {
        const float dj = density[lo[k].idx], w = lo[k].w;
        densityTemp += dj * w;
        ps += (di + dj) * w;
    }
    float s = 0.f;
    s += densityTemp;  // score[i] += densityTemp happens at step i, before any higher index scatters into it
    for (int k = off_up[i]; k < off_up[i + 1]; k++) s += density[up[k].idx] * up[k].w;
    score[i] = s;
    pair_sum[i] = ps;
}

__global__ __launch_bounds__(256) void density_update(const float *__restrict__ density, const float *__restrict__ score,
                                                      const float *__restrict__ normalizer, int N, float *__restrict__ density_out,
                                                      double *__restrict__ chg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float nd = score[i] * normalizer[0];
    if (nd > 2.f) nd = 2.f;
    const float df = density[i] - nd;
    chg[i] = (double)(df * df);
    density_out[i] = nd;
}

// second level of the global sums: the 256 chunk sums added in order by one thread, then the scalar the loop needs --
// the normaliser (float)(N / sum) of heuristic.cpp:118, or the mean squared change of heuristic.cpp:133
__global__ void finish_normalizer(const double *__restrict__ v256, int N, float *__restrict__ normalizer)
{
    double sum = 0.;
    for (int c = 0; c < 256; c++) sum += v256[c];
    normalizer[0] = (float)(N / sum);
}
__global__ void finish_change(const double *__restrict__ v256, int N, double *__restrict__ change)
{
    double sum = 0.;
    for (int c = 0; c < 256; c++) sum += v256[c];
    change[0] = sum / N;
}

// fixed-shape first level of the global sums: chunk c = the sum of a contiguous range of ceil(N / 256) elements, formed by a
// reduction tree whose shape depends on N only -- thread t of the chunk's workgroup adds elements t, t + 256, ... in that
// order, then the 256 partial sums are folded pairwise (stride 128, 64, ... 1).  Deterministic run to run and rank to rank.
// (The oracle adds everything sequentially in double; the two meet again when N / sum is rounded to float, heuristic.cpp:118.)
// History: one thread per chunk walking it left to right took ~2 ms per call at 2 M points, a wavefront adding in index
// order through v_readlane 340 us; the tree takes a few microseconds.
__global__ __launch_bounds__(256) void chunk_sums(const double *__restrict__ v, int N, double *__restrict__ out256)
{
    __shared__ double part[256];
    const int c = blockIdx.x, t = threadIdx.x;
    const int per = (N + 255) / 256;
    const int s = c * per, e = min(s + per, N);
    double acc = 0.;
    for (int i = s + t; i < e; i += 256) acc += v[i];
    part[t] = acc;
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
        if (t < stride) part[t] += part[t + stride];
        __syncthreads();
    }
    if (t == 0) out256[c] = part[0];
}

// ---- greedy selection (heuristic.cpp:139-163) on the device -----------------------------------------------------------
// The reference walks the points by descending density; a point whose score is still >= 0.7 is kept and lowers the score
// of its lower-index neighbours.  That looks serial but the decision of point j depends only on the kept points i with
// j in lower(i) -- the list up(j) -- that come BEFORE j in the order.  So: sort once (radix sort of
// (~density bits, index) keys = density descending, index ascending: the reference's stable order), then decide in
// rounds -- a point is decided as soon as all its earlier-ranked up-neighbours are, applying their subtractions in rank
// order with the reference's arithmetic ((float)(score - (double)density * w)).  Exactly the sequential result; the
// number of rounds is the longest dependency chain.
__global__ __launch_bounds__(256) void greedy_keys(const float *__restrict__ density, int N, unsigned long long *__restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float d = density[i];
    const unsigned hi = (d != d) ? 0u : ~__builtin_bit_cast(unsigned, d);  // densities are >= +0; NaN (no neighbours at all) first, by index
    keys[i] = ((unsigned long long)hi << 32) | (unsigned)i;
}

__global__ __launch_bounds__(256) void greedy_ranks(const unsigned long long *__restrict__ sorted, int N, int *__restrict__ rank)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < N) rank[(unsigned)sorted[p]] = p;
}

// state: 0 undecided, 1 kept, 2 dropped
__global__ __launch_bounds__(256) void greedy_round(unsigned char *__restrict__ state, const int *__restrict__ rank,
                                                    const float *__restrict__ density, const float *__restrict__ score0,
                                                    const int *__restrict__ off_up, const Nb *__restrict__ up, int N, float limit,
                                                    int *__restrict__ undecided)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N || state[j] != 0) return;
    const int rj = rank[j], k0 = off_up[j], k1 = off_up[j + 1];
    for (int k = k0; k < k1; k++) {
        const int i = up[k].idx;
        if (rank[i] < rj && state[i] == 0) {
            atomicAdd(undecided, 1);
            return;
        }
    }
    float s = score0[j];
    int last = -1;
    for (;;) {  // kept earlier neighbours in rank order (lists are short: selection by repeated minimum)
        int best = 0x7fffffff, bk = -1;
        for (int k = k0; k < k1; k++) {
            const int i = up[k].idx, ri = rank[i];
            if (ri < rj && ri > last && ri < best && state[i] == 1) {
                best = ri;
                bk = k;
            }
        }
        if (bk < 0) break;
        s = (float)((double)s - (double)density[up[bk].idx] * (double)up[bk].w);
        last = best;
    }
    state[j] = (s < limit) ? 2 : 1;
}

// Segmented ordering of the upper lists by ONE global radix sort: key = (owner point << 32) | sub-key, value = the entry
// itself.  Segment boundaries (off_up) are unchanged because the owner is the major key.  Used when the lists are long
// (dense clouds): the per-thread insertion sort (sort_upper) and the repeated-minimum selection in greedy_round are
// O(L^2) per point -- 64 ms and 365 ms at 60 k points with ~100 neighbours each, against ~1 ms for the two sorts.
__global__ __launch_bounds__(256) void list_keys(const int *__restrict__ off_up, const Nb *__restrict__ up, int N,
                                               const int *__restrict__ rank /* nullable: sub-key = index */,
                                               unsigned long long *__restrict__ keys)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    for (int k = off_up[j]; k < off_up[j + 1]; k++) {
        const int i = up[k].idx;
        keys[k] = ((unsigned long long)(unsigned)j << 32) | (unsigned)(rank ? rank[i] : i);
    }
}

// greedy_round on lists sorted by the RANK of the neighbour: the earlier-ranked neighbours are a prefix, readiness is
// checked from its end (the latest-ranked one is the likeliest to be undecided), the subtractions are one forward pass
__global__ __launch_bounds__(256) void greedy_round_sorted(unsigned char *__restrict__ state, const int *__restrict__ rank,
                                                           const float *__restrict__ density, const float *__restrict__ score0,
                                                           const int *__restrict__ off_up, const Nb *__restrict__ up_by_rank, int N,
                                                           float limit, int *__restrict__ undecided)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N || state[j] != 0) return;
    const int rj = rank[j], k0 = off_up[j], k1 = off_up[j + 1];
    int lo = k0, hi = k1;  // first entry whose rank is not earlier than rj
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (rank[up_by_rank[mid].idx] < rj)
            lo = mid + 1;
        else
            hi = mid;
    }
    for (int k = lo - 1; k >= k0; k--)
        if (state[up_by_rank[k].idx] == 0) {
            atomicAdd(undecided, 1);
            return;
        }
    float s = score0[j];
    for (int k = k0; k < lo; k++)
        if (state[up_by_rank[k].idx] == 1) s = (float)((double)s - (double)density[up_by_rank[k].idx] * (double)up_by_rank[k].w);
    state[j] = (s < limit) ? 2 : 1;
}

}  // namespace mvs

using namespace mvs;

extern "C" {

int mvs_filter_points(mvs_ctx *ctx, const float *points4, int npoints, float alpha, int32_t *keep_out, int *out_count)
{
    if (!ctx || !out_count || npoints < 0 || (npoints > 0 && (!points4 || !keep_out)))
        return fail(ctx, MVS_EINVAL, "mvs_filter_points: bad arguments");
    *out_count = 0;
    if (npoints == 0) return MVS_OK;
    const float radius = alpha / 4.f;  // heuristic.cpp:63
    if (!(radius > 0.f) || !std::isfinite(radius)) return fail(ctx, MVS_EINVAL, "mvs_filter_points: alpha must be positive and finite");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const int N = npoints;
    hipStream_t st = ctx->stream;
    // MVS_FILTER_TIMING=1: wall time of each stage on stderr (adds a stream synchronisation per stage)
    const bool timing = ctx->hooks.filter_timing;
    auto clock_now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = clock_now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const auto t = clock_now();
        fprintf(stderr, "filter_points[%d] %-28s %8.2f ms\n", N, what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    unsigned table = 1;
    while (table < 2u * (unsigned)N) table <<= 1;
    const unsigned mask = table - 1;
    const float cell = std::sqrt(radius);

    // pass 1 buffers
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_p4 = al(sizeof(float) * 4 * N), b_p3 = al(sizeof(float) * 3 * N), b_c3 = al(sizeof(int) * 3 * N), b_head = al(sizeof(int) * table),
                 b_n1 = al(sizeof(int) * ((size_t)N + 1));
    int rc = ensure(ctx, ctx->flow_arena, b_p4 + b_p3 + b_c3 + b_head + 6 * b_n1 + 4 * al(sizeof(float) * N) + 2 * al(sizeof(double) * N) + 8192);
    if (rc) return rc;
    char *base = (char *)ctx->flow_arena.ptr;
    float *d_p4 = (float *)base;
    float *d_p3 = (float *)(base += b_p4);
    int *d_c3 = (int *)(base += b_p3);
    int *d_head = (int *)(base += b_c3);
    int *d_next = (int *)(base += b_head);
    int *d_cnt_lo = (int *)(base += b_n1), *d_cnt_up = (int *)(base += b_n1), *d_off_lo = (int *)(base += b_n1), *d_off_up = (int *)(base += b_n1),
        *d_fill = (int *)(base += b_n1);
    float *d_density = (float *)(base += b_n1), *d_score = (float *)(base += al(sizeof(float) * N));
    float *d_density_b = (float *)(base += al(sizeof(float) * N)), *d_score_b = (float *)(base += al(sizeof(float) * N));
    double *d_pair = (double *)(base += al(sizeof(float) * N)), *d_chg = (double *)(base += al(sizeof(double) * N));
    double *d_256 = (double *)(base += al(sizeof(double) * N));

    MVS_HIP(ctx, hipMemcpyAsync(d_p4, points4, sizeof(float) * 4 * N, hipMemcpyHostToDevice, st));
    MVS_HIP(ctx, hipMemsetAsync(d_head, 0xff, sizeof(int) * table, st));
    MVS_HIP(ctx, hipMemsetAsync(d_cnt_up, 0, sizeof(int) * ((size_t)N + 1), st));
    MVS_HIP(ctx, hipMemsetAsync(d_cnt_lo, 0, sizeof(int) * ((size_t)N + 1), st));
    MVS_HIP(ctx, hipMemsetAsync(d_fill, 0, sizeof(int) * ((size_t)N + 1), st));
    const unsigned g = (unsigned)div_up(N, 256);
    dehomog_cells<<<g, 256, 0, st>>>(d_p4, N, 1.0f / cell, d_p3, d_c3);
    hash_build<<<g, 256, 0, st>>>(d_c3, N, mask, d_head, d_next);
    nb_count<<<g, 256, 0, st>>>(d_p3, d_c3, d_head, d_next, mask, N, radius, d_cnt_lo, d_cnt_up);
    {
        // offsets = exclusive prefix sums of the counts over N + 1 entries (count[N] = 0, so entry N is the total); integer
        // sums are exact under any algorithm.  The single-workgroup scan this replaces took 5.7 ms per call at 2 M points.
        size_t scan_bytes = 0;
        if (rocprim::exclusive_scan(nullptr, scan_bytes, d_cnt_lo, d_off_lo, 0, (size_t)N + 1, rocprim::plus<int>(), st) != hipSuccess)
            return fail(ctx, MVS_EHIP, "mvs_filter_points: scan sizing failed");
        if ((rc = ensure(ctx, ctx->r_tmp0, scan_bytes > 0 ? scan_bytes : 1))) return rc;
        if (rocprim::exclusive_scan(ctx->r_tmp0.ptr, scan_bytes, d_cnt_lo, d_off_lo, 0, (size_t)N + 1, rocprim::plus<int>(), st) != hipSuccess ||
            rocprim::exclusive_scan(ctx->r_tmp0.ptr, scan_bytes, d_cnt_up, d_off_up, 0, (size_t)N + 1, rocprim::plus<int>(), st) != hipSuccess)
            return fail(ctx, MVS_EHIP, "mvs_filter_points: scan failed");
    }
    // a 32-bit prefix sum that wraps shows as an offset smaller than its predecessor (counts are >= 0): flag in the spare entry of d_fill
    offsets_wrap_check<<<g, 256, 0, st>>>(d_off_lo, d_off_up, N, d_fill + N);
    MVS_HIP(ctx, hipGetLastError());
    int total = 0, wrapped = 0;
    MVS_HIP(ctx, hipMemcpyAsync(&total, d_off_lo + N, sizeof(int), hipMemcpyDeviceToHost, st));
    MVS_HIP(ctx, hipMemcpyAsync(&wrapped, d_fill + N, sizeof(int), hipMemcpyDeviceToHost, st));
    MVS_HIP(ctx, hipStreamSynchronize(st));
    if (total < 0 || wrapped)
        return fail(ctx, MVS_ENOMEM, "mvs_filter_points: neighbour table overflows 2^31 entries (radius %g is far beyond the spacing of %d points)", (double)radius, N);
    lap("upload, hash, count, scan");

    // neighbour lists live in their own buffer (size known only now)
    if ((rc = ensure(ctx, ctx->r_tmp2, 2 * sizeof(Nb) * (size_t)(total > 0 ? total : 1)))) return rc;
    Nb *d_lo = (Nb *)ctx->r_tmp2.ptr, *d_up = d_lo + (total > 0 ? total : 1);
    // long lists (dense clouds): order them with global radix sorts instead of per-thread O(L^2) loops (see list_keys)
    const int force_sorted = ctx->hooks.filter_sorted_lists;  // test hook: 1 always, 0 never
    const bool long_lists = (force_sorted >= 0 ? force_sorted == 1 : (long long)total > 24ll * N) && total > 0;
    if (long_lists)
        nb_fill<false><<<g, 256, 0, st>>>(d_p3, d_c3, d_head, d_next, mask, N, radius, d_off_lo, d_off_up, d_fill, d_lo, d_up);
    else
        nb_fill<true><<<g, 256, 0, st>>>(d_p3, d_c3, d_head, d_next, mask, N, radius, d_off_lo, d_off_up, d_fill, d_lo, d_up);
    unsigned long long *d_skeys = nullptr, *d_skeys2 = nullptr;
    Nb *d_up_alt = nullptr;
    size_t pair_bytes = 0;
    int key_bits = 33;
    while (key_bits < 64 && (1ull << (key_bits - 32)) < (unsigned long long)N) key_bits++;
    if (long_lists) {
        if ((rc = ensure(ctx, ctx->filter_sort, 3 * sizeof(unsigned long long) * (size_t)total))) return rc;
        d_skeys = (unsigned long long *)ctx->filter_sort.ptr;
        d_skeys2 = d_skeys + total;
        d_up_alt = (Nb *)(d_skeys2 + total);
        if (rocprim::radix_sort_pairs(nullptr, pair_bytes, d_skeys, d_skeys2, (unsigned long long *)d_up, (unsigned long long *)d_up_alt,
                                      (size_t)total, 0u, (unsigned)key_bits, st) != hipSuccess)
            return fail(ctx, MVS_EHIP, "mvs_filter_points: list sort sizing failed");
        if ((rc = ensure(ctx, ctx->r_tmp0, pair_bytes > 0 ? pair_bytes : 1))) return rc;
        // lower lists by index (the order heuristic.cpp:74-92 leaves them in), sorted into the spare region and copied back
        list_keys<<<g, 256, 0, st>>>(d_off_lo, d_lo, N, nullptr, d_skeys);
        if (rocprim::radix_sort_pairs(ctx->r_tmp0.ptr, pair_bytes, d_skeys, d_skeys2, (unsigned long long *)d_lo, (unsigned long long *)d_up_alt,
                                      (size_t)total, 0u, (unsigned)key_bits, st) != hipSuccess)
            return fail(ctx, MVS_EHIP, "mvs_filter_points: list sort failed");
        MVS_HIP(ctx, hipMemcpyAsync(d_lo, d_up_alt, sizeof(Nb) * (size_t)total, hipMemcpyDeviceToDevice, st));
        // upper lists by index
        list_keys<<<g, 256, 0, st>>>(d_off_up, d_up, N, nullptr, d_skeys);
        if (rocprim::radix_sort_pairs(ctx->r_tmp0.ptr, pair_bytes, d_skeys, d_skeys2, (unsigned long long *)d_up, (unsigned long long *)d_up_alt,
                                      (size_t)total, 0u, (unsigned)key_bits, st) != hipSuccess)
            return fail(ctx, MVS_EHIP, "mvs_filter_points: list sort failed");
        std::swap(d_up, d_up_alt);  // d_up: by index (what the power iteration gathers in); d_up_alt: free again
    } else {
        sort_upper<<<g, 256, 0, st>>>(d_off_up, N, d_up);
    }
    MVS_HIP(ctx, hipGetLastError());

    lap("neighbour fill + sort");
    // power iteration (heuristic.cpp:103-136)
    std::vector<float> ones((size_t)N, 1.f);
    MVS_HIP(ctx, hipMemcpyAsync(d_density, ones.data(), sizeof(float) * N, hipMemcpyHostToDevice, st));
    // Iteration k reads dens[k & 1], writes scor[k & 1] and dens[(k + 1) & 1]; both reductions finish on the device and
    // only the 8-byte mean squared change comes back.  Iteration k + 1 is queued before the host looks at iteration k's
    // value, so the GPU never waits for the host; when the loop should have stopped at k, the speculative iteration has
    // only touched the OTHER score / density buffers and is simply ignored (same result as the one-at-a-time loop).
    double *d_256b = d_256 + 256;
    float *d_norm = (float *)(d_256 + 512);
    double *d_change = d_256 + 520;
    if (!ctx->filter_pinned) {
        MVS_HIP(ctx, hipHostMalloc((void **)&ctx->filter_pinned, 64, hipHostMallocDefault));
        for (int e = 0; e < 2; e++) MVS_HIP(ctx, hipEventCreateWithFlags(&ctx->filter_ev[e], hipEventDisableTiming));
    }
    double *h_change = ctx->filter_pinned;
    float *dens[2] = {d_density, d_density_b}, *scor[2] = {d_score, d_score_b};
    auto launch_iteration = [&](int k) {
        density_score<<<g, 256, 0, st>>>(dens[k & 1], d_off_lo, d_lo, d_off_up, d_up, N, scor[k & 1], d_pair);
        chunk_sums<<<256, 256, 0, st>>>(d_pair, N, d_256);
        finish_normalizer<<<1, 1, 0, st>>>(d_256, N, d_norm);
        density_update<<<g, 256, 0, st>>>(dens[k & 1], scor[k & 1], d_norm, N, dens[(k + 1) & 1], d_chg);
        chunk_sums<<<256, 256, 0, st>>>(d_chg, N, d_256b);
        finish_change<<<1, 1, 0, st>>>(d_256b, N, d_change + (k & 1));
        (void)hipMemcpyAsync(h_change + (k & 1), d_change + (k & 1), sizeof(double), hipMemcpyDeviceToHost, st);
        (void)hipEventRecord(ctx->filter_ev[k & 1], st);
    };
    int it = 0, k = 0;
    launch_iteration(0);
    for (;;) {
        if (k + 1 < 200) launch_iteration(k + 1);
        MVS_HIP(ctx, hipEventSynchronize(ctx->filter_ev[k & 1]));
        it = k + 1;
        if (!(h_change[k & 1] > 1e-6 && it < 200)) break;
        k++;
    }
    MVS_HIP(ctx, hipGetLastError());
    d_density = dens[(k + 1) & 1];  // the density after the last counted update, and the score it was computed from
    d_score = scor[k & 1];
    if (timing) fprintf(stderr, "filter_points[%d] power iterations: %d, neighbour pairs: %d\n", N, it, total);
    lap("power iteration");

    // greedy selection (heuristic.cpp:139-163): order on the device, decisions in dependency rounds (see greedy_round)
    unsigned long long *d_keys = (unsigned long long *)d_pair, *d_sorted = (unsigned long long *)d_chg;  // the f64 scratch of the power iteration
    int *d_rank = d_fill;                                                                               // free since nb_fill
    unsigned char *d_state = (unsigned char *)d_cnt_lo;                                                 // free since the scans
    int *d_undecided = d_cnt_up;                                                                        // 8 counters, one per round of a batch
    greedy_keys<<<g, 256, 0, st>>>(d_density, N, d_keys);
    size_t cub_bytes = 0;
    if (rocprim::radix_sort_keys(nullptr, cub_bytes, d_keys, d_sorted, (size_t)N, 0u, 64u, st) != hipSuccess)
        return fail(ctx, MVS_EHIP, "mvs_filter_points: radix sort sizing failed");
    if ((rc = ensure(ctx, ctx->r_tmp0, cub_bytes > 0 ? cub_bytes : 1))) return rc;
    if (rocprim::radix_sort_keys(ctx->r_tmp0.ptr, cub_bytes, d_keys, d_sorted, (size_t)N, 0u, 64u, st) != hipSuccess)
        return fail(ctx, MVS_EHIP, "mvs_filter_points: radix sort failed");
    greedy_ranks<<<g, 256, 0, st>>>(d_sorted, N, d_rank);
    MVS_HIP(ctx, hipMemsetAsync(d_state, 0, (size_t)N, st));
    if (d_up_alt) {  // long lists: a second ordering of the upper lists, by the rank of the neighbour
        list_keys<<<g, 256, 0, st>>>(d_off_up, d_up, N, d_rank, d_skeys);
        if ((rc = ensure(ctx, ctx->r_tmp0, pair_bytes > 0 ? pair_bytes : 1))) return rc;  // the key sort above may have grown it
        if (rocprim::radix_sort_pairs(ctx->r_tmp0.ptr, pair_bytes, d_skeys, d_skeys2, (unsigned long long *)d_up, (unsigned long long *)d_up_alt,
                                      (size_t)total, 0u, (unsigned)key_bits, st) != hipSuccess)
            return fail(ctx, MVS_EHIP, "mvs_filter_points: rank sort failed");
    }
    MVS_HIP(ctx, hipGetLastError());
    lap("device sort by density");
    const float densityLimit = .7f;
    const int max_rounds = ctx->hooks.filter_max_rounds;  // 2048 unless the test hook says otherwise
    int rounds = 0;
    for (bool done = false; !done;) {
        int h_und[8];
        MVS_HIP(ctx, hipMemsetAsync(d_undecided, 0, sizeof(h_und), st));
        for (int r = 0; r < 8; r++)
            if (d_up_alt)
                greedy_round_sorted<<<g, 256, 0, st>>>(d_state, d_rank, d_density, d_score, d_off_up, d_up_alt, N, densityLimit, d_undecided + r);
            else
                greedy_round<<<g, 256, 0, st>>>(d_state, d_rank, d_density, d_score, d_off_up, d_up, N, densityLimit, d_undecided + r);
        MVS_HIP(ctx, hipGetLastError());
        MVS_HIP(ctx, hipMemcpyAsync(h_und, d_undecided, sizeof(h_und), hipMemcpyDeviceToHost, st));
        MVS_HIP(ctx, hipStreamSynchronize(st));
        rounds += 8;
        for (int r = 0; r < 8; r++) done = done || h_und[r] == 0;
        if (!done && rounds >= max_rounds) break;
    }
    if (timing) fprintf(stderr, "filter_points[%d] greedy rounds (batches of 8): %d\n", N, rounds);
    std::vector<unsigned char> state((size_t)N);
    MVS_HIP(ctx, hipMemcpyAsync(state.data(), d_state, (size_t)N, hipMemcpyDeviceToHost, st));
    MVS_HIP(ctx, hipStreamSynchronize(st));
    bool undecided_left = false;
    for (int i = 0; i < N && !undecided_left; i++) undecided_left = state[i] == 0;
    if (undecided_left) {
        // The number of rounds is the longest dependency chain: tens for a surface-like cloud, but up to N for a
        // pathological one (points strung along a line with monotone density).  Past max_rounds the rest is decided by the
        // reference's sequential walk on the host, continuing from the decisions already made -- the same result, since a
        // decided point is final and an undecided one only ever waits for earlier-ranked points.
        std::vector<float> density((size_t)N), score((size_t)N);
        std::vector<int> off((size_t)N + 1);
        std::vector<Nb> lo((size_t)(total > 0 ? total : 1));
        std::vector<unsigned long long> sorted((size_t)N);
        MVS_HIP(ctx, hipMemcpyAsync(density.data(), d_density, sizeof(float) * N, hipMemcpyDeviceToHost, st));
        MVS_HIP(ctx, hipMemcpyAsync(score.data(), d_score, sizeof(float) * N, hipMemcpyDeviceToHost, st));
        MVS_HIP(ctx, hipMemcpyAsync(off.data(), d_off_lo, sizeof(int) * ((size_t)N + 1), hipMemcpyDeviceToHost, st));
        if (total > 0) MVS_HIP(ctx, hipMemcpyAsync(lo.data(), d_lo, sizeof(Nb) * (size_t)total, hipMemcpyDeviceToHost, st));
        MVS_HIP(ctx, hipMemcpyAsync(sorted.data(), d_sorted, sizeof(unsigned long long) * N, hipMemcpyDeviceToHost, st));
        MVS_HIP(ctx, hipStreamSynchronize(st));
        for (int p = 0; p < N; p++) {  // heuristic.cpp:146-163 in rank order; scores are rebuilt from the start, so every kept
            const int ord = (int)(unsigned)sorted[p];  // point (decided on the device or here) applies its subtractions once
            const bool keep_it = state[ord] == 1 || (state[ord] == 0 && !(score[ord] < densityLimit));
            if (state[ord] == 0) state[ord] = keep_it ? 1 : 2;
            if (!keep_it) continue;
            const double localDensity = density[ord];
            for (int k = off[ord]; k < off[ord + 1]; k++) score[lo[k].idx] = (float)(score[lo[k].idx] - localDensity * lo[k].w);
        }
        if (timing) fprintf(stderr, "filter_points[%d] finished on the host after %d rounds\n", N, rounds);
    }
    int m = 0;
    for (int i = 0; i < N; i++)
        if (state[i] == 1) keep_out[m++] = i;
    *out_count = m;
    lap("device greedy rounds + output");
    return MVS_OK;
}

}  // extern "C"
