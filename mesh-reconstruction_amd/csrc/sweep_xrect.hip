// sweep_xrect.hip -- the EXACT-f32 sampler's plane sweep (arithmetic contract v1, sweep.hip / DESIGN.md section 2a: the f32 restatement of
// shader.frag:13-24 + the RGB8 read-back of render_glx.cpp:359) for RECTIFIED side views on gfx950: the same cells as sweep_tiled, bit
// for bit, at about a third of its vector instructions.
//
// A side view is rectified against the main view when its matrix Q has q1 = q4 = q8 = q9 = q10 = 0 (csrc/sweep_rect.hip: a pure
// translation inside the main camera's focal plane with equal intrinsics -- the fronto-parallel sweep and SURVEY.md 8d's ring).  Then,
// in the contract's own arithmetic (every vanishing term left where it stands: fma(0, t, u) = u exactly),
//     cx = fma(z, q2, fma(q0, xn, q3)) * r      depends on (column, plane, view) only
//     cy = fma(z, q6, fma(q5, yn, q7)) * r      depends on (row, plane, view) only,      r = RN(1 / q11) per view
// so the projection, the reciprocal, trunc / fract and the in-frame test -- 11 of sweep_tiled's 16.8 vector instructions per sample --
// are done ONCE per (column, plane, view) for the 8 rows of a tile and once per (row, plane, view) for its 64 columns.  Unlike the fixed
// sampler's rectified kernel nothing has to be certified: the sub-texel fractions stay per-lane (ax) and per-row (ay) values, computed by
// the contract's expressions themselves; only the WORK is shared, no value is assumed.  What is left per sample is the texture fetch:
//     h    = LDS[slot + (iy_j - y0) * RS + (ix - x0)]                         one ds_read_b64 (f16 quad {t00 + 1/2, dxt, dyt, dxy})
//     res  = fma(ay_j, fma(ax, dxy, dyt), fma(ax, dxt, t00 + 1/2));  Iq = (int)res;  cell += |Iq - I_main|       6 vector instructions
//
// Thread mapping and region pipeline are sweep_fx_rect's: workgroup = one 64 x 8 tile x one 16-plane chunk, a wavefront owns all 8 rows
// and 4 of the 16 planes (lane = column), two LDS slots, the next (chunk, view) box copied by buffer_load_dwordx4 ... lds while the
// current one is sampled, one s_waitcnt vmcnt(0) + s_barrier per region.  The boxes are separable (x range per tile column, y range per
// tile row) and EXACT: cx is a monotone function of the column and of z (compositions of roundings of monotone functions), so the four
// corner evaluations of plan_xrect bound every sample of the tile -- there is no slow path; a plan whose boxes do not fit the slots
// falls back to sweep_tiled as a whole.
#include "sweep_shared.hpp"

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace mvs {

namespace {

constexpr int XR_TILE_H = 8, XR_PC = 16, XR_KW = 4;  // tile rows, planes per chunk, planes per wavefront (round 5 built 8 planes per wavefront on 32-plane chunks -- the code below is
                                                      // generic in both -- and dropped it: on a ring of views the widest and the tallest box belong to DIFFERENT views, the slot holds their product,
                                                      // and at c3 it no longer fits 24 KiB: the plan falls back to sweep_tiled; where it fits, the copy volume per sample does not shrink.
                                                      // Also built in round 5 and not kept: 64 x 16 tiles with EIGHT wavefronts of 2 planes (a plane's set-up serving 16 rows, a third less
                                                      // copy volume): 16 rows of Im / best / bi make 150 VGPRs -- at the 128 cap that 512-thread workgroups need for two per CU the compiler
                                                      // spills 44 of them (88 bytes of scratch per lane); at 168 one workgroup per CU is left)
constexpr int XR_MAX_NI = 6;                          // 1 KiB copy instructions per wavefront and region, at most (24 KiB per slot: c3's boxes are ~84 quads x 24 rows of 8 bytes)
constexpr int XR_MAX_REGIONS = 256;                    // (chunks x views) of one workgroup: 16 KiB of records at the most
constexpr int XR_WAVES = 4;                           // launch bound: <= 128 VGPRs (two 16 KiB slots + 16 KiB of records at c3: three workgroups per CU)

typedef const __attribute__((address_space(4))) uint32_t *cu32;
typedef const __attribute__((address_space(4))) float *cf32;
template <typename T, typename U>
__device__ __forceinline__ T as_const(const U *p) { return (T)(uintptr_t)p; }

struct XrArgs {
    const uint2 *__restrict__ quads16;   // f16 quad images of the side views, pad_slab quads each
    const uint32_t *__restrict__ xrec;   // [tiles_x][V][NC][8]  X records (plan_xrect)
    const uint32_t *__restrict__ yrec;   // [tiles_y][V][NC][8]  Y records
    const float *__restrict__ Q;         // V x 12
    const float *__restrict__ z;         // D
    const uint8_t *__restrict__ main_img;
    uint32_t *__restrict__ volume;
    float *__restrict__ depth;
    float *__restrict__ cost;
    int *__restrict__ index;
    uint2 *__restrict__ part;            // plane-split launches: partial bests [gridDim.y][P]
    size_t pad_slab;
    int pitch, W, H, D, V, v0, vcount, nchunks, chunk0, chunk1, cps, ty0, tyn, tiles_x;
    int rs;          // quads per LDS row (even)
    int slot_bytes;  // bytes per LDS slot (a multiple of 4096)
    float invW, invH, Wp, Hp;
};

__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t acc)
{
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000);
}

// ------------------------------------------------------------------------------------------------------
// planner: one 8-dword record per (tile column, view, chunk) and per (tile row, view, chunk)
// ------------------------------------------------------------------------------------------------------
// X record: q0, q2, q3, r = RN(1 / q11) | box (x0 | quads << 16; 0 quads: nothing of the tile column in frame) | full | 8 x0 | some
// Y record: q5, q6, q7, 8 (pad_slab v + y0 pitch) | box (y0 | rows << 16) | full | 0 | some
// full: bit d = every valid pixel of the tile column (row) is in frame at plane d of the chunk; some: bit d = some pixel is.
// cx over the tile's columns and the chunk's planes is monotone in both (compositions of roundings of monotone functions): the box is
// bounded by the four corner evaluations, and a plane is in frame for every column iff it is for the two end columns.  Samples out of
// frame (cx <= 0.5 or >= W + 0.5) are masked in the kernel and need no texel, hence the clamp.  stats: [0] widest box, [1] tallest.
__global__ __launch_bounds__(256) void plan_xrect(XrArgs a, int tiles_y, uint32_t *__restrict__ xrec, uint32_t *__restrict__ yrec, int *__restrict__ stats)
{
    const int NC = a.nchunks;
    const int nx = a.tiles_x * a.V * NC, ny = tiles_y * a.V * NC;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = tid < nx + ny;
    const bool is_y = tid >= nx;
    const int e = live ? (is_y ? tid - nx : tid) : 0;
    const int chunk = e % NC, v = (e / NC) % a.V, t = e / (NC * a.V);
    const float *q = a.Q + 12 * v;
    const float r = rcp_rn(q[11]);
    const int d0 = chunk * XR_PC;
    const int size = is_y ? a.H : a.W;
    const int p0 = t * (is_y ? XR_TILE_H : TILE_W), p1 = min(p0 + (is_y ? XR_TILE_H : TILE_W), size) - 1;
    const float lim = (float)size + 0.5f;
    float lo = 3.0e38f, hi = -3.0e38f;
    uint32_t flags = 0u, some = 0u;
    for (int k = 0; k < XR_PC; k++) {
        const float z = a.z[min(d0 + k, a.D - 1)];
        float c[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int pix = s ? p1 : p0;
            if (is_y) {
                const float yn = __builtin_fmaf(-(float)(2 * pix + 1), a.invH, 1.0f);
                c[s] = __builtin_fmaf(z, q[6], __builtin_fmaf(q[5], yn, q[7])) * r;
            } else {
                const float xn = __builtin_fmaf((float)(2 * pix + 1), a.invW, -1.0f);
                c[s] = __builtin_fmaf(z, q[2], __builtin_fmaf(q[0], xn, q[3])) * r;
            }
        }
        const float cl = fminf(c[0], c[1]), ch = fmaxf(c[0], c[1]);
        if (cl > 0.5f && ch < lim) flags |= 1u << k;
        if (ch > 0.5f && cl < lim) some |= 1u << k;
        if (d0 + k < a.D) {
            lo = fminf(lo, cl);
            hi = fmaxf(hi, ch);
        }
    }
    uint32_t box = 0u;
    int extent = 0, i0 = 0;
    if (hi > 0.5f && lo < lim && lo == lo && hi == hi) {  // something can be in frame: the texels of the in-frame samples, i = trunc(c) in [0, size]
        i0 = (int)fmaxf(lo, 0.0f);
        int i1 = (int)fminf(hi, lim);
        i0 = max(0, min(i0, size));
        i1 = max(i0, min(i1, size));
        if (!is_y) i0 &= ~1;  // 16-byte copy units = two quads
        extent = i1 - i0 + 1;
        if (!is_y) extent = (extent + 1) & ~1;
        box = (uint32_t)i0 | ((uint32_t)extent << 16);
    } else {
        flags = some = 0u;
    }
    if (live) {
        uint32_t *rec = (is_y ? yrec : xrec) + (size_t)e * 8;
        if (is_y) {
            rec[0] = __builtin_bit_cast(uint32_t, q[5]);
            rec[1] = __builtin_bit_cast(uint32_t, q[6]);
            rec[2] = __builtin_bit_cast(uint32_t, q[7]);
            rec[3] = 8u * ((uint32_t)a.pad_slab * (uint32_t)v + (uint32_t)i0 * (uint32_t)a.pitch);
        } else {
            rec[0] = __builtin_bit_cast(uint32_t, q[0]);
            rec[1] = __builtin_bit_cast(uint32_t, q[2]);
            rec[2] = __builtin_bit_cast(uint32_t, q[3]);
            rec[3] = __builtin_bit_cast(uint32_t, r);
        }
        rec[4] = box;
        rec[5] = flags;
        rec[6] = is_y ? 0u : 8u * (uint32_t)i0;
        rec[7] = some;
    }
    // one atomic per wavefront and counter (the host waits for the kernel before it reads them)
    int wx = (live && !is_y) ? extent : 0, wy = (live && is_y) ? extent : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        wx = max(wx, __shfl_xor(wx, m, 64));
        wy = max(wy, __shfl_xor(wy, m, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (wx) atomicMax(stats, wx);
        if (wy) atomicMax(stats + 1, wy);
    }
}

// ------------------------------------------------------------------------------------------------------
// sweep kernel
// ------------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The 64-byte record of a region out of LDS.  Inline asm: the compiler orders a ds_read it can see behind ALL pending LDS copies; issue
// and wait sit in ONE statement -- between two statements the compiler is free to move a loaded register before its data has arrived.
__device__ __forceinline__ void lds_read_record(uint32_t addr, u32x4 &x0, u32x4 &x1, u32x4 &y0, u32x4 &y1)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1)
                 : "v"(addr));
}

// RS: the LDS row stride in quads as a compile-time constant (the planner rounds the widest box up to one of a few values), so that the 8
// row reads of a plane take immediate offsets instead of one v_add_u32 each; 0: run-time stride (boxes wider than the largest constant)
template <bool WRITE_VOLUME, bool FUSED, int RS>
__global__ __launch_bounds__(256, XR_WAVES) void sweep_exact_rect(XrArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)smem;

    const int band_tile = grouped_tile(blockIdx.x, a.tiles_x, a.tyn);
    if (band_tile < 0) return;
    const int tx = band_tile % a.tiles_x, ty = band_tile / a.tiles_x + a.ty0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = tx * TILE_W + lane;
    const int row0 = ty * XR_TILE_H;
    const bool col_ok = col < a.W;
    const int NC = a.nchunks;
    const int nrows = min(XR_TILE_H, a.H - row0);
    const float xn = __builtin_fmaf((float)(2 * col + 1), a.invW, -1.0f);
    const int rowl = row0 + (lane & 7);  // lane l also works for row l & 7 of the tile (per-row values are computed once, on 8 lanes' worth of work)
    const float ynl = __builtin_fmaf(-(float)(2 * rowl + 1), a.invH, 1.0f);

    uint32_t Im[8];
#pragma unroll
    for (int j = 0; j < 8; j++) Im[j] = (col_ok && row0 + j < a.H) ? (uint32_t)a.main_img[(size_t)(row0 + j) * a.W + col] : 0u;

    const int chunk_first = a.chunk0 + (int)blockIdx.y * a.cps;
    const int chunk_last = min(a.chunk1, chunk_first + a.cps);
    const int vend = a.v0 + a.vcount;
    const int nreg = (chunk_last - chunk_first) * a.vcount;

    // The records of ALL this workgroup's regions go into LDS once (16 dwords per region: X record, Y record): read from there a
    // region's constants cost an LDS round trip instead of dependent scalar loads from memory on the critical path of every region
    // (the first form of this kernel spent 0.74 of its 1.66 ms at c3 on that skeleton).
    const uint32_t ctab_byte = 2u * (uint32_t)a.slot_bytes;
    {
        uint32_t *ctab = smem + (ctab_byte >> 2);
        const uint32_t *xrec = a.xrec + (size_t)tx * a.V * NC * 8, *yrec = a.yrec + (size_t)ty * a.V * NC * 8;
        for (int i = threadIdx.x; i < nreg * 16; i += 256) {
            const int e = i >> 4, w = i & 15;
            const int chunk = chunk_first + e / a.vcount, v = a.v0 + e % a.vcount;
            ctab[i] = w < 8 ? xrec[(size_t)(v * NC + chunk) * 8 + w] : yrec[(size_t)(v * NC + chunk) * 8 + (w - 8)];
        }
    }
    __syncthreads();

    // per-lane source offsets (bytes) of this wavefront's copy instructions: instruction i = wave + 4 t fills LDS bytes [1024 i, 1024 i + 1024)
    // of the slot = 16-byte units g = 64 i + lane of the dense [row][rs] region image
    const int units = a.rs >> 1;
    uint32_t srcoff[XR_MAX_NI];
#pragma unroll
    for (int t = 0; t < XR_MAX_NI; t++) {
        const int g = (wave + 4 * t) * 64 + lane;
        srcoff[t] = 8u * (uint32_t)((g / units) * a.pitch + (g % units) * 2);
    }
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(a.quads16, 0xffffffffu);
    const uint32_t rs8 = 8u * (uint32_t)a.rs;

    // copy of the region whose X / Y records are (xr, yr) into the slot at `slot_byte`
    auto issue_copy = [&](const u32x4 &xr1, const u32x4 &yr0, const u32x4 &yr1, uint32_t slot_byte) {
        const uint32_t xb = (uint32_t)__builtin_amdgcn_readfirstlane((int)xr1.x), yb = (uint32_t)__builtin_amdgcn_readfirstlane((int)yr1.x);
        if ((xb >> 16) == 0u || (yb >> 16) == 0u) return;
        const int n = (int)(yb >> 16) * units;  // whole rows of rs quads (what lies right of the box is copied along and never read)
        const uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane((int)xr1.z) + (uint32_t)__builtin_amdgcn_readfirstlane((int)yr0.w);
        char *dst = (char *)smem + slot_byte + wave * 1024;
#pragma unroll
        for (int t = 0; t < XR_MAX_NI; t++) {
            int left = n - (wave + 4 * t) * 64;  // units of the region from this instruction's first one on (wave-uniform)
            asm volatile("" : "+s"(left));
            if (left > 0) {
                if (left >= 64 || lane < left)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void *)(dst + t * 4096), 16, srcoff[t], src, 0, 0);
            }
        }
    };

    uint32_t acc[8][XR_KW];
    uint32_t best[8];
    int bi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        best[j] = 1u;  // (sum 1, count 0): the cross-multiplied comparison is true for the first cell with a view in frame, false for empty ones
        bi[j] = -1;
#pragma unroll
        for (int k = 0; k < XR_KW; k++) acc[j][k] = 0u;
    }
    int notfull[XR_KW];  // per plane of this wavefront: views of the current chunk whose count did NOT go to every cell of the plane (wave-uniform)
    float zc[XR_KW];
#pragma unroll
    for (int k = 0; k < XR_KW; k++) {
        notfull[k] = 0;
        zc[k] = 0.0f;
    }

    uint32_t slot_cur = 0u, slot_nxt = (uint32_t)a.slot_bytes;
    int chunk = chunk_first, v = a.v0;
    const uint32_t ctab_addr = lds_base + ctab_byte;
    // 32 bytes per wavefront behind the records: the 8 row fractions of a plane go from lanes 0..7 to ALL lanes through LDS -- one ds_write_b32,
    // then one ds_read_b32 per row with the same address on every lane (a broadcast: conflict-free, and issued to the LDS pipe, which this
    // kernel leaves two thirds idle) -- instead of one v_readlane_b32 per (row, plane) on the vector unit, which is what bounds it (round 5:
    // tools/valu_microbench prices v_readlane at 4.2 of the row loop's 27 cycles)
    const uint32_t fy_wave_addr = ctab_addr + 64u * (uint32_t)max(nreg, 1) + 32u * (uint32_t)wave;
    const uint32_t fy_lane_addr = fy_wave_addr + 4u * (uint32_t)(lane & 7);
    // records of region 0 (its copy goes out before the loop) -- 64 bytes per region
    u32x4 x0r, x1r, y0r, y1r;
    lds_read_record(ctab_addr, x0r, x1r, y0r, y1r);
    if (nreg > 0) issue_copy(x1r, y0r, y1r, slot_cur);

    for (int r = 0; r < nreg; r++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's copies of region r have landed ...
        __builtin_amdgcn_s_barrier();                      // ... and everybody else's; nobody reads region r - 1 any more
        // the records of region r + 1 (past the last region: of the last one again), its copy into the other slot: in flight during this region's sampling
        const uint32_t nxt = ctab_addr + 64u * (uint32_t)min(r + 1, nreg - 1);
        u32x4 nx0, nx1, ny0, ny1;
        lds_read_record(nxt, nx0, nx1, ny0, ny1);
        if (r + 1 < nreg) issue_copy(nx1, ny0, ny1, slot_nxt);

        if (v == a.v0) {  // first view of a chunk: this wavefront's four planes
            const cf32 zs = as_const<cf32>(a.z);
#pragma unroll
            for (int k = 0; k < XR_KW; k++) zc[k] = zs[min(chunk * XR_PC + wave * XR_KW + k, a.D - 1)];
        }

        // ---- sample region r ----
        // (through integer temporaries: __builtin_bit_cast applied directly to an ext-vector element reads element 0 with this hipcc)
        const uint32_t u0 = x0r.x, u1 = x0r.y, u2 = x0r.z, u3 = x0r.w, u5 = y0r.x, u6 = y0r.y, u7 = y0r.z;
        const float q0 = __builtin_bit_cast(float, u0), q2 = __builtin_bit_cast(float, u1), q3 = __builtin_bit_cast(float, u2), rr = __builtin_bit_cast(float, u3);
        const float q5 = __builtin_bit_cast(float, u5), q6 = __builtin_bit_cast(float, u6), q7 = __builtin_bit_cast(float, u7);
        const uint32_t xb = (uint32_t)__builtin_amdgcn_readfirstlane((int)x1r.x), yb = (uint32_t)__builtin_amdgcn_readfirstlane((int)y1r.x);
        const uint32_t xfl = (uint32_t)__builtin_amdgcn_readfirstlane((int)x1r.y), yfl = (uint32_t)__builtin_amdgcn_readfirstlane((int)y1r.y);
        const uint32_t xsm = (uint32_t)__builtin_amdgcn_readfirstlane((int)x1r.w), ysm = (uint32_t)__builtin_amdgcn_readfirstlane((int)y1r.w);
        const float Ax = __builtin_fmaf(q0, xn, q3), Ayl = __builtin_fmaf(q5, ynl, q7);
        const int x0 = (int)(xb & 0xffffu), y0 = (int)(yb & 0xffffu);
        const uint32_t slot_addr = lds_base + slot_cur;
        const uint32_t both_fl = (xfl & yfl) >> (XR_KW * wave);  // bit k: plane k of this wavefront is in frame for every pixel of the tile
        const uint32_t both_sm = (xsm & ysm) >> (XR_KW * wave);  // bit k: for some pixel
#pragma unroll
        for (int k = 0; k < XR_KW; k++) {
            if (!((both_sm >> k) & 1u)) {  // nothing of the tile in frame at this plane
                notfull[k]++;
                continue;
            }
            const float cx = __builtin_fmaf(zc[k], q2, Ax) * rr;
            const float cyl = __builtin_fmaf(zc[k], q6, Ayl) * rr;
            const float fx = __builtin_amdgcn_fractf(cx);
            const uint32_t addrx = slot_addr + (uint32_t)(((int)cx - x0) << 3);
            const float fyl = __builtin_amdgcn_fractf(cyl);
            const int iyl = (int)cyl;
            const int iy0 = __builtin_amdgcn_readlane(iyl, 0);
            // rows of the tile usually sample consecutive texel rows (cy advances by one per row up to its rounding): then a row's address is the previous one + a stride
            const bool consecutive = ((uint32_t)__builtin_amdgcn_ballot_w64(iyl - (lane & 7) == iy0) & 0xffu) == 0xffu;
            if (((both_fl >> k) & 1u) && consecutive) {
                // the row fractions to LDS, the 8 quad reads and the 8 broadcast reads of the fractions, one wait, then the arithmetic
                unsigned long long h[8];
                float fyv[8];
                uint32_t ad = addrx + (uint32_t)(iy0 - y0) * rs8;
                asm volatile("ds_write_b32 %0, %1" ::"v"(fy_lane_addr), "v"(fyl) : "memory");
                if (RS > 0) {
#pragma unroll
                    for (int j = 0; j < 8; j++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(h[j]) : "v"(ad), "n"(j * (RS > 0 ? RS : 1) * 8));
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        asm volatile("ds_read_b64 %0, %1" : "=v"(h[j]) : "v"(ad));
                        ad += rs8;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fyv[j]) : "v"(fy_wave_addr), "n"(4 * j));
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), "+v"(fyv[0]), "+v"(fyv[1]), "+v"(fyv[2]),
                               "+v"(fyv[3]), "+v"(fyv[4]), "+v"(fyv[5]), "+v"(fyv[6]), "+v"(fyv[7]));
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const half4_t q4 = __builtin_bit_cast(half4_t, h[j]);
                    const float fy = fyv[j];
                    const float ta = __builtin_fmaf(fx, (float)q4[1], (float)q4[0]);
                    const float tb = __builtin_fmaf(fx, (float)q4[3], (float)q4[2]);
                    acc[j][k] = sad_u32((uint32_t)(int)__builtin_fmaf(fy, tb, ta), Im[j], acc[j][k]);
                }
            } else {
                // a tile at the border of the side view, or rows that do not advance in step: the frame tests per lane and per row, each
                // sampled cell counting its sample itself
                notfull[k]++;
                const bool inx = col_ok && cx > 0.5f && cx < a.Wp;
                const uint32_t ym = (uint32_t)__builtin_amdgcn_ballot_w64(rowl < a.H && cyl > 0.5f && cyl < a.Hp) & 0xffu;
                // the rows' values out of lanes 0..7 BEFORE the lanes out of frame are masked off (a v_readlane inside the divergent
                // region would read what the compiler computed there -- for the active lanes only)
                const uint32_t rowoffl = (uint32_t)(iyl - y0) * rs8;
                uint32_t ro[8];
                float fyr[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    ro[j] = (uint32_t)__builtin_amdgcn_readlane((int)rowoffl, j);
                    fyr[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fyl), j));
                }
                if (inx) {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        if ((ym >> j) & 1u) {  // wave-uniform
                            const uint32_t ad = addrx + ro[j];
                            unsigned long long hq;
                            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(hq) : "v"(ad));
                            const half4_t q4 = __builtin_bit_cast(half4_t, hq);
                            const float fy = fyr[j];
                            const float ta = __builtin_fmaf(fx, (float)q4[1], (float)q4[0]);
                            const float tb = __builtin_fmaf(fx, (float)q4[3], (float)q4[2]);
                            acc[j][k] = sad_u32((uint32_t)(int)__builtin_fmaf(fy, tb, ta), Im[j], acc[j][k] + 65536u);
                        }
                    }
                }
            }
        }

        // ---- chunk epilogue ----
        if (v + 1 == vend) {
            const int d0 = chunk * XR_PC + wave * XR_KW;
            const size_t P = (size_t)a.W * a.H;
            const uint32_t pix0 = 4u * (uint32_t)(row0 * a.W + col);  // byte offset of this lane's first pixel inside a plane
            if (col_ok) {
#pragma unroll
                for (int k = 0; k < XR_KW; k++) {
                    const uint32_t cntk = (uint32_t)(a.vcount - notfull[k]) << 16;  // the views whose count goes to every cell of the plane
                    if (d0 + k < a.D) {
                        // one resource per plane (a volume can exceed the 4 GiB a resource spans), rows by the wave-uniform offset
                        const __amdgpu_buffer_rsrc_t rvol = make_rsrc(WRITE_VOLUME ? a.volume + (size_t)(d0 + k) * P : nullptr, 0xffffffffu);
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            if (j < nrows) {
                                const uint32_t cell = acc[j][k] + cntk;
                                if (WRITE_VOLUME) __builtin_amdgcn_raw_buffer_store_b32(cell, rvol, pix0, 4u * (uint32_t)(j * a.W), 2);  // nt: written once, read by a later kernel
                                if (FUSED) {
                                    const bool better = umul24u(cell & 0xffffu, best[j] >> 16) < umul24u(best[j] & 0xffffu, cell >> 16);
                                    best[j] = better ? cell : best[j];
                                    bi[j] = better ? d0 + k : bi[j];
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < XR_KW; k++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j][k] = 0u;
#pragma unroll
            for (int k = 0; k < XR_KW; k++) notfull[k] = 0;
        }

        if (++v == vend) {
            v = a.v0;
            chunk++;
        }
        x0r = nx0;
        x1r = nx1;
        y0r = ny0;
        y1r = ny1;
        const uint32_t sw = slot_cur;
        slot_cur = slot_nxt;
        slot_nxt = sw;
    }

    // ---- depth selection across the four wavefronts (each holds the best of its own planes): lowest cost, ties -> lowest plane ----
    if (FUSED) {
        __syncthreads();  // every copy has landed and every sample loop is done: the slots are free
        uint2 *ex = (uint2 *)smem;  // [wave][row][lane]
#pragma unroll
        for (int j = 0; j < 8; j++) ex[(wave * 8 + j) * 64 + lane] = make_uint2(best[j], (uint32_t)bi[j]);
        __syncthreads();
        const size_t P = (size_t)a.W * a.H;
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const int j = wave * 2 + jj;
            const int row = row0 + j;
            if (col_ok && row < a.H) {
                uint32_t b = 1u;
                int bidx = -1;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const uint2 c = ex[(s * 8 + j) * 64 + lane];
                    if ((int)c.y >= 0) {
                        const uint32_t lhs = umul24u(c.x & 0xffffu, b >> 16), rhs = umul24u(b & 0xffffu, c.x >> 16);
                        const bool take = lhs < rhs || (lhs == rhs && bidx >= 0 && (int)c.y < bidx);
                        b = take ? c.x : b;
                        bidx = take ? (int)c.y : bidx;
                    }
                }
                if (bidx < 0) b = 0u;  // no plane had a view in frame: the empty cell, as argmin_update_packed leaves it
                const size_t pix = (size_t)row * a.W + col;
                if (a.part) {
                    a.part[(size_t)blockIdx.y * P + pix] = make_uint2(b, (uint32_t)bidx);
                } else {  // store_best<CS_EXACT>
                    a.depth[pix] = bidx >= 0 ? a.z[bidx] : MVS_BACKGROUND_DEPTH;
                    a.cost[pix] = bidx >= 0 ? cell_cost<CS_EXACT>(b & 0xffffu, b >> 16) : __builtin_inff();
                    a.index[pix] = bidx;
                }
            }
        }
    }
}

void fill_args(mvs_ctx *ctx, const SweepParams &p, XrArgs &a)
{
    memset(&a, 0, sizeof(a));
    a.quads16 = p.quads16;
    a.Q = p.Q;
    a.z = p.z;
    a.main_img = p.main_img;
    a.volume = p.volume;
    a.depth = p.depth;
    a.cost = p.cost;
    a.index = p.index;
    a.part = nullptr;
    a.pad_slab = p.pad_slab;
    a.pitch = p.pitch;
    a.W = p.W;
    a.H = p.H;
    a.D = p.D;
    a.V = p.V;
    a.v0 = p.v0;
    a.vcount = p.vcount;
    a.nchunks = div_up(p.D, XR_PC);
    a.chunk0 = p.chunk0;
    a.chunk1 = p.chunk1;
    a.ty0 = p.ty0;
    a.tyn = p.tyn;
    a.tiles_x = div_up(p.W, TILE_W);
    a.invW = p.invW;
    a.invH = p.invH;
    a.Wp = p.Wp;
    a.Hp = p.Hp;
    a.rs = ctx->xrect_rs;
    a.slot_bytes = ctx->xrect_slot_bytes;
    const size_t nxb = (size_t)a.tiles_x * a.V * a.nchunks;
    a.xrec = (const uint32_t *)ctx->xrect_tab.ptr + 16;
    a.yrec = a.xrec + 8 * nxb;
}

}  // namespace

bool rect_view_host(const float *q);  // sweep_rect.hip

// Decides whether the exact sampler's rectified kernel serves the current (views, planes) -- every view rectified, every box within
// the LDS slots -- and builds its box tables (one small planner launch, one read-back of two counters).  Sets ctx->xrect_ok.
int sweep_xrect_plan(mvs_ctx *ctx)
{
    ctx->xrect_ok = false;
    if (ctx->hooks.no_rect || ctx->V == 0 || ctx->V > XR_MAX_REGIONS) return MVS_OK;
    for (int v = 0; v < ctx->V; v++)
        if (!rect_view_host(ctx->q_host.data() + 12 * v)) return MVS_OK;
    if ((unsigned long long)ctx->pad_slab * (unsigned long long)ctx->V * 8ull >= (1ull << 32)) return MVS_OK;  // one buffer resource, 32-bit byte offsets
    SweepParams p;
    fill_params(ctx, p, 0, ctx->V, XR_TILE_H, XR_PC);
    ctx->xrect_rs = 2;
    ctx->xrect_slot_bytes = 4096;
    XrArgs a;
    fill_args(ctx, p, a);
    const int tiles_y = div_up(ctx->H, XR_TILE_H);
    const size_t nxb = (size_t)a.tiles_x * a.V * a.nchunks, nyb = (size_t)tiles_y * a.V * a.nchunks;
    int rc;
    if ((rc = ensure(ctx, ctx->xrect_tab, (16 + 8 * (nxb + nyb)) * sizeof(uint32_t)))) return rc;
    fill_args(ctx, p, a);  // (the table may have moved)
    int *stats = (int *)ctx->xrect_tab.ptr;
    MVS_HIP(ctx, hipMemsetAsync(stats, 0, 64, ctx->stream));
    plan_xrect<<<(unsigned)((nxb + nyb + 255) / 256), 256, 0, ctx->stream>>>(a, tiles_y, (uint32_t *)a.xrec, (uint32_t *)a.yrec, stats);
    MVS_HIP(ctx, hipGetLastError());
    int h[2] = {0, 0};
    MVS_HIP(ctx, hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int max_rw = h[0], max_rh = h[1];
    if (ctx->hooks.rect_verbose) {
        fprintf(stderr, "sweep_xrect_plan: widest box %d quads, tallest %d rows\n", max_rw, max_rh);
        uint32_t rec[32];
        for (int which = 0; which < 2; which++) {
            if (hipMemcpy(rec, which ? a.yrec : a.xrec, sizeof(rec), hipMemcpyDeviceToHost) != hipSuccess) break;
            for (int e = 0; e < 4; e++)
                fprintf(stderr, "  %c record %d: %g %g %g %08x | box %08x flags %08x src %u\n", which ? 'Y' : 'X', e, __builtin_bit_cast(float, rec[8 * e]), __builtin_bit_cast(float, rec[8 * e + 1]),
                        __builtin_bit_cast(float, rec[8 * e + 2]), rec[8 * e + 3], rec[8 * e + 4], rec[8 * e + 5], rec[8 * e + 6]);
        }
    }
    if (max_rw <= 0 || max_rh <= 0) return MVS_OK;  // nothing in frame anywhere: the general kernel writes the empty cells
    int rs = (max_rw + 1) & ~1;
    for (int fixed : {72, 84, 96, 112, 128})  // the strides sweep_exact_rect is instantiated for (immediate row offsets); wider boxes: run-time stride
        if (rs <= fixed) {
            rs = fixed;
            break;
        }
    const int units = rs / 2;
    const int instrs = div_up(max_rh * units, 64);  // 1 KiB copy instructions per region
    if (instrs > 4 * XR_MAX_NI) return MVS_OK;      // wide baselines / few planes: boxes too large for the slots
    ctx->xrect_rs = rs;
    ctx->xrect_slot_bytes = div_up(instrs, 4) * 4096;
    ctx->xrect_ok = true;
    return MVS_OK;
}

// launch of the rectified exact-sampler sweep; `p` carries the plane / row / view ranges.  Returns the split count like sweep_fx_launch.
int sweep_xrect_launch(mvs_ctx *ctx, SweepParams &p, bool vol, bool fused, unsigned flags)
{
    XrArgs a;
    fill_args(ctx, p, a);
    const int groups = div_up(a.tiles_x, 2) * div_up(p.tyn, 4);
    const int nch = p.chunk1 - p.chunk0, tiles = a.tiles_x * p.tyn;
    int want = (int)((flags >> 16) & 0xffu);
    if (!want) want = div_up(16 * ctx->num_cus, tiles);
    p.cps = div_up(nch, max(1, min(want, nch)));
    p.cps = max(1, min(p.cps, XR_MAX_REGIONS / max(1, p.vcount)));  // the records of a workgroup's regions live in LDS (64 bytes each)
    size_t lds = 2 * (size_t)a.slot_bytes + 64 * (size_t)p.cps * (size_t)max(1, p.vcount) + 128;  // slots, records, 32 bytes per wavefront for the row fractions
    if (fused) lds = lds < 16384 ? 16384 : lds;  // the cross-wavefront depth selection borrows 16 KiB
    a.cps = p.cps;
    const int nsplit = div_up(nch, p.cps);
    int rc;
    if (fused && nsplit > 1) {
        if ((rc = ensure(ctx, ctx->best_parts, (size_t)nsplit * ctx->W * ctx->H * sizeof(uint2)))) return rc;
        p.part = (uint2 *)ctx->best_parts.ptr;
        a.part = p.part;
    }
    const dim3 grid((unsigned)(div_up(groups, 8) * 64), (unsigned)nsplit);
    auto go = [&](auto kernel) -> int {
        MVS_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kernel<<<grid, 256, lds, ctx->stream>>>(a);
        MVS_HIP(ctx, hipGetLastError());
        return MVS_OK;
    };
    auto pick = [&](auto rs_tag) -> int {
        constexpr int RS = decltype(rs_tag)::value;
        if (vol && fused) return go(sweep_exact_rect<true, true, RS>);
        if (vol) return go(sweep_exact_rect<true, false, RS>);
        return go(sweep_exact_rect<false, true, RS>);
    };
    switch (a.rs) {
    case 72: rc = pick(std::integral_constant<int, 72>{}); break;
    case 84: rc = pick(std::integral_constant<int, 84>{}); break;
    case 96: rc = pick(std::integral_constant<int, 96>{}); break;
    case 112: rc = pick(std::integral_constant<int, 112>{}); break;
    case 128: rc = pick(std::integral_constant<int, 128>{}); break;
    default: rc = pick(std::integral_constant<int, 0>{}); break;
    }
    if (rc) return rc;
    return nsplit;
}

}  // namespace mvs
