// sweep_rect.hip -- the fixed sampler's plane sweep (arithmetic contract v2, sweep_fx.hip / DESIGN.md section 2b) for RECTIFIED side
// views on gfx950: the same cells, bit for bit, at a third of the vector instructions.
//
// A side view is "rectified" against the main view when its matrix Q (csrc/camera_math.cpp) has q1 = q4 = q8 = q9 = q10 = 0: the
// side camera is a pure translation of the main one inside its focal plane with the same intrinsics -- the classic fronto-parallel
// plane sweep, and SURVEY.md section 8d's benchmark ring.  Then, in the contract's own f32 arithmetic,
//     Tx = fma(fma(z, bx, fma(q0, xn, q3)), 256 r, magic + 4)      depends on (column, plane, view) only
//     Ty = fma(fma(z, by, fma(q5, yn, q7)), 256 r, magic + 4)      depends on (row, plane, view) only,  r = RN(1 / q11) per view
// and in exact arithmetic both advance by exactly 256 per pixel: every pixel of a plane samples the side view at the SAME sub-texel
// phase (kx, ky) and at the texel its own position plus one common shift.  The f32 roundings break that on a few per cent of the
// (plane, view) pairs (a coordinate within ~0.07 / 256 texel of a rounding boundary).  So a planner kernel EVALUATES the contract's
// expression for every column of a 64-column tile (and every row of an 8-row tile) and records, per (tile column, view, plane) and
// per (tile row, view, plane), the first pixel's integer texel and phase plus two certificates: "every pixel of the tile is in frame"
// and "every pixel has the view's nominal phase and the texel of the first pixel plus its own offset".  Where both hold -- 96-98 % of
// the (wavefront, plane, view) triples -- the sweep kernel needs NO per-lane coordinate arithmetic at all:
//     quad   = LDS[slot + (iy - y0 + j) * RS + (ix - x0) + lane]          one ds_read_b32 at a wave-uniform base + lane
//     cell  += |v_dot4_u32_u8(quad, W[ky][kx]) - 255 I_main|               weight word in an SGPR; v_dot4 + v_sad_u16 = 2 VALU / sample
// (sweep_fx_tiled: 9.25 in its sample loop, 12.4 overall).  Where a certificate fails, that plane of that wavefront evaluates the
// general expression per lane with texels gathered from the view's quad image in global memory (`slow_plane`): the same function of
// the same inputs, so the two kernels agree on every cell (tests/test_sweep_gpu.py: rect vs general vs oracle).
//
// Thread mapping.  Workgroup = 256 threads = one 64 x 8-pixel tile x one 16-plane chunk, as in sweep_fx_tiled (same planner boxes);
// but a wavefront owns ALL 8 rows of the tile and 4 of the 16 planes (lane = column): 32 accumulators per thread, and a plane's
// uniform work (decode, weight word, LDS base) is shared by 8 rows.
//
// Region pipeline.  Per (chunk, view) the planner's box of the view's quad image is copied into one of two LDS slots by
// buffer_load_dwordx4 ... lds (dense rows of RS quads, 1 KiB = 64 lanes x 16 B per instruction, per-lane source offsets fixed for the
// launch), one region ahead of the one being sampled; s_waitcnt vmcnt(0) + one s_barrier per region make every wavefront's part
// visible and free the slot the next copy goes to.  The X / Y records of the coming regions travel through vector registers (lane l =
// dword l, v_readlane when the region's turn comes).  The sample loop is one inline-asm statement per plane (M0-based
// ds_read_addtid_b32, v_dot4_u32_u8, v_sad_u16; planes that are not FULL run it with EXEC = 0): the compiler would order every ds_read
// behind ALL pending LDS copies, and a per-plane branch costs more scalar instructions than the plane's arithmetic.
// What bounds it (profiles/r03): vector-instruction issue.  Measured and rejected: a second region of look-ahead
// with counted waits (three slots, records through an LDS ring; +10 %), row-wise copies (+8 %), copies that all hit in L2 (no change).
#include "sweep_shared.hpp"

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace mvs {

namespace {

constexpr float RX_MAGIC = 12582912.0f;  // 1.5 * 2^23 (FX_MAGIC of sweep_fx.hip)
constexpr int RX_TILE_H = 8, RX_PC = 16, RX_KW = 4;  // tile rows, planes per chunk, planes per wavefront
constexpr int RX_MAX_NI = 3;                          // copy instructions per wavefront and region, at most
#ifndef RX_DOUBLE_BUFFER
#define RX_DOUBLE_BUFFER 0
#endif
#ifndef RX_WAVES
#define RX_WAVES 5
#endif
constexpr int RX_WAVES_PER_SIMD = RX_WAVES;                  // launch bound: <= 128 VGPRs

enum RxMode : unsigned { RX_SKIP = 0, RX_FAST = 1, RX_BORDER = 2, RX_GENERIC = 3 };  // = FxMode (the plan is plan_regions_fx's)

constexpr uint32_t RX_UNIFORM = 1u << 30;

// Read-only tables are read through the CONSTANT address space: a wave-uniform load from it is a scalar load (s_load_dword*)
// whatever the kernel stores elsewhere (through a global pointer the compiler must assume the volume stores may alias it and
// falls back to vector loads after the first store).
typedef const __attribute__((address_space(4))) uint32_t *cu32;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x2 *cu2;
typedef const __attribute__((address_space(4))) u32x4 *cu4;
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) u32x8 *cu8;
typedef const __attribute__((address_space(4))) float *cf32;
template <typename T, typename U>
__device__ __forceinline__ T as_const(const U *p) { return (T)(uintptr_t)p; }
constexpr int RX_BIAS = 1 << 15;  // table entries hold t0 + RX_BIAS: the extrapolated texel of a tile's pixel 0 may lie left of / above the image

__device__ __forceinline__ uint32_t sad_u16(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u16(a, b, acc); }

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most n of this wavefront's vector-memory operations are outstanding (n wave-uniform)
__device__ __forceinline__ void wait_vmcnt(int n)
{
    // a smaller count than asked for is always safe (it waits for more)
    switch (n) {
    case 1: wait_vm<1>(); break;
    case 2: wait_vm<2>(); break;
    case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;
    case 6: wait_vm<6>(); break;
    case 7: wait_vm<6>(); break;
    case 8: wait_vm<8>(); break;
    case 9: wait_vm<9>(); break;
    case 10: case 11: wait_vm<10>(); break;
    case 12: case 13: case 14: wait_vm<12>(); break;
    case 15: case 16: case 17: wait_vm<15>(); break;
    case 18: wait_vm<18>(); break;
    default: wait_vm<0>(); break;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// planner: per-axis tables
// ------------------------------------------------------------------------------------------------------
// A view is eligible when Tx does not depend on the row, Ty not on the column and w on nothing (and w > 0).
template <typename Q>
__device__ __forceinline__ bool rect_view(Q q) { return q[1] == 0.0f && q[4] == 0.0f && q[8] == 0.0f && q[9] == 0.0f && q[10] == 0.0f && q[11] > 0.0f; }

// 256 r of an eligible view, as sweep_fx_tiled forms it: the w row divided by 256 (exact), s.w = fma(z, 0, q11 / 256) = q11 / 256
template <typename Q>
__device__ __forceinline__ float rect_r256(Q q) { return rcp_rn(q[11] * 0.00390625f); }

// Tx of column c / Ty of row r at plane z: the contract's expression with the terms that vanish left in (fma(0, t, u) = u exactly)
template <typename Q>
__device__ __forceinline__ float rect_tx(Q q, float r256, float z, int c, float invW)
{
    const float xn = __builtin_fmaf((float)(2 * c + 1), invW, -1.0f);
    const float ax = __builtin_fmaf(q[0], xn, __builtin_fmaf(q[1], 0.0f, q[3]));
    return __builtin_fmaf(__builtin_fmaf(z, q[2], ax), r256, RX_MAGIC + 4.0f);
}
template <typename Q>
__device__ __forceinline__ float rect_ty(Q q, float r256, float z, int r, float invH)
{
    const float yn = __builtin_fmaf(-(float)(2 * r + 1), invH, 1.0f);
    const float ay = __builtin_fmaf(q[4], 0.0f, __builtin_fmaf(q[5], yn, q[7]));
    return __builtin_fmaf(__builtin_fmaf(z, q[6], ay), r256, RX_MAGIC + 4.0f);
}

// Table set of one plan (device pointers into ctx->rect_tab).  T = tiles_x resp. tiles_y, NC = plane chunks, dpad = 16 NC.
struct RectTables {
    uint32_t *xt, *yt;    // [T][V][dpad]   full entries (below)
    uint32_t *xmm, *ymm;  // [T][V][dpad]   lowest | highest << 16 integer texel over the tile's in-frame pixels, 0xffffffff if none
    uint32_t *wt;         // [V][dpad]      weight word of the view's nominal phase pair at that plane
    uint32_t *xw;         // [tiles_x][V][NC][4 wavefronts][8]  X records (pass C)
    uint32_t *yr;         // [tiles_y][V][NC][4 wavefronts][4]  Y records
    uint32_t *xbox, *ybox;  // [T][V][NC]   region boxes (pass B)
    int *stats;           // [0] widest region (quads), [1] tallest region (rows), [2] planes that are not FULL (diagnostic)
    int dpad;
};

// Pass A.  One thread per (tile column or tile row, view, plane), evaluating every pixel of the tile.  Full entry:
//   bits  0-19  t0 + RX_BIAS, t0 = (u >> 3) - 32 i of the in-frame pixels (i = index in the tile, u = 1/256-texel coordinate):
//               phase in the low 5 bits, integer texel of the tile's pixel 0 (extrapolated, possibly left of the image) above
//   bits 20-26  n = pixels of the tile that are out of frame (all of the tile's pixels: 64 resp. 8)
//   bit  27     which end: 0 = the first n pixels are out, 1 = the last n
//   bit  30     certificate: the out-of-frame pixels are exactly those n, every other pixel has (u >> 3) - 32 i == t0, and
//               the phase is the view's nominal one for this plane (the phase at the image centre: the W table's)
// Pixels past the image edge (ragged last tile) count as matching.
__global__ __launch_bounds__(256) void plan_rect_axis(SweepParams p, RectTables rt, const uint32_t *__restrict__ lut)
{
    const int dpad = rt.dpad;
    const int nx = p.tiles_x * p.V * dpad, ny = p.tiles_y * p.V * dpad, nw = p.V * dpad;
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nx + ny + nw) return;
    const int which = tid < nx ? 0 : (tid < nx + ny ? 1 : 2);
    if (which == 1) tid -= nx;
    if (which == 2) tid -= nx + ny;
    const int d = tid % dpad, v = (tid / dpad) % p.V, t = tid / (dpad * p.V);
    const float *q = p.Q + 12 * v;
    const bool elig = rect_view(q);
    const float r256 = rect_r256(q);
    const float z = p.z[min(d, p.D - 1)];
    const float hix = RX_MAGIC + 132.0f + 256.0f * (float)p.W, hiy = RX_MAGIC + 132.0f + 256.0f * (float)p.H;
    const uint32_t nomx = (__builtin_bit_cast(uint32_t, rect_tx(q, r256, z, p.W / 2, p.invW)) >> 3) & 31u;
    const uint32_t nomy = (__builtin_bit_cast(uint32_t, rect_ty(q, r256, z, p.H / 2, p.invH)) >> 3) & 31u;
    if (which == 2) {
        rt.wt[tid] = lut[nomy * 32u + nomx];
        return;
    }
    const int size = which == 0 ? TILE_W : RX_TILE_H;
    const int first = t * size;
    const int count = min(first + size, which == 0 ? p.W : p.H) - first;  // pixels of the tile inside the image
    int nin = 0, first_in = -1, last_in = -1;
    int t0 = 0;
    uint32_t lo = 0xffffu, hi = 0u;
    bool uniform = elig;
    for (int i = 0; i < count; i++) {
        const float T = which == 0 ? rect_tx(q, r256, z, first + i, p.invW) : rect_ty(q, r256, z, first + i, p.invH);
        if (!(T > RX_MAGIC + 132.0f && T < (which == 0 ? hix : hiy))) continue;
        const uint32_t u = __builtin_bit_cast(uint32_t, T) & 0x3fffffu;
        const int ti = (int)(u >> 3) - 32 * i;
        if (nin == 0) {
            t0 = ti;
            first_in = i;
        }
        uniform = uniform && ti == t0;
        lo = min(lo, u >> 8);
        hi = max(hi, u >> 8);
        last_in = i;
        nin++;
    }
    int nout = size, side = 0;
    if (nin > 0) {
        uniform = uniform && last_in - first_in + 1 == nin;             // contiguous
        if (first_in == 0) {                                            // the tail is out (or nothing: pixels past the image edge are don't-cares)
            side = 1;
            nout = last_in == count - 1 ? 0 : size - 1 - last_in;
        } else {
            side = 0;
            nout = first_in;
            uniform = uniform && last_in == count - 1;                  // cut at both ends: not representable
        }
        uniform = uniform && (uint32_t)(t0 & 31) == (which == 0 ? nomx : nomy) && t0 + RX_BIAS >= 0 && t0 + RX_BIAS < (1 << 20);
    }
    // (no pixel in frame: n = the whole tile and the certificate as computed so far -- nothing to sample)
    (which == 0 ? rt.xt : rt.yt)[tid] = ((uint32_t)(t0 + RX_BIAS) & 0xfffffu) | ((uint32_t)nout << 20) | ((uint32_t)side << 27) | (uniform ? RX_UNIFORM : 0u);
    (which == 0 ? rt.xmm : rt.ymm)[tid] = nin > 0 ? (lo | (hi << 16)) : 0xffffffffu;
}

// Pass B.  One thread per (tile column or tile row, view, chunk): the box of the view's quad image the chunk's 16 planes touch
// (x origin a multiple of 4 quads: the copies move 16-byte units):  box = origin | extent << 16 | any << 31
// (any: some pixel of some plane is in frame).  Counters: widest / tallest box.
__global__ __launch_bounds__(256) void plan_rect_box(SweepParams p, RectTables rt)
{
    const int NC = p.nchunks, dpad = rt.dpad;
    const int nx = p.tiles_x * p.V * NC, ny = p.tiles_y * p.V * NC;
    const bool live = (int)(blockIdx.x * blockDim.x + threadIdx.x) < nx + ny;  // (no early return: the counters are reduced per wavefront)
    int tid = min((int)(blockIdx.x * blockDim.x + threadIdx.x), nx + ny - 1);
    const int which = tid < nx ? 0 : 1;
    if (which == 1) tid -= nx;
    const int chunk = tid % NC, v = (tid / NC) % p.V, t = tid / (NC * p.V);
    const uint32_t *mm = (which == 0 ? rt.xmm : rt.ymm) + ((size_t)t * p.V + v) * dpad + chunk * RX_PC;
    uint32_t lo = 0xffffu, hi = 0u;
    bool any = false;
    for (int k = 0; k < RX_PC; k++) {
        const uint32_t m = mm[k];
        if (m == 0xffffffffu) continue;
        any = true;
        lo = min(lo, m & 0xffffu);
        hi = max(hi, m >> 16);
    }
    const uint32_t org = any ? (which == 0 ? (lo & ~3u) : lo) : 0u;
    const uint32_t ext = any ? hi + 1u - org : 0u;
    // widest / tallest box: one atomic per wavefront (21 000 threads on two addresses took 0.23 ms)
    const int wx = wave_max_i32(live && any && which == 0 ? (int)ext : 0), wy = wave_max_i32(live && any && which == 1 ? (int)ext : 0);
    if ((threadIdx.x & 63) == 0) {
        if (wx) atomicMax(rt.stats + 0, wx);
        if (wy) atomicMax(rt.stats + 1, wy);
    }
    if (live) (which == 0 ? rt.xbox : rt.ybox)[tid] = org | (min(ext, 0x7fffu) << 16) | (any ? 1u << 31 : 0u);
}

// Pass C (after the host has chosen the row stride RS of the LDS slots from pass B's counters).  One thread per (tile column or
// tile row, view, chunk, wavefront): that wavefront's record of the region.  Per plane a 16-bit field with the plane's share of the
// LDS byte offset of the texel quad of the tile's pixel 0 -- x: 4 (ix - x0 + RX_BIAS_X); y: 4 RS (iy - y0 + RX_BIAS_Y); the biases
// because pixel 0 of a tile that is partly out of frame lies left of / above the box (the kernel's copies land RX_BIAS bytes into
// the slot, so base + x share + y share is the quad's address; the sum stays below 2^14) -- or a flag (x: bit 14, y: bit 15) = the
// certificate failed, look at the full entry.  A plane whose certificate holds but whose tile is partly (MASKED) or wholly (NONE) out
// of frame has no flag: its out-of-frame pixels come as a mask byte per plane (x: n | side << 7 as in the full entry, n = 64: nothing
// in frame; y: the bit mask of the rows that are OUT of frame), zero for FULL planes.
//   X record (8 dwords): x01, x23, W0, W1, W2, W3, 4 x0 | (any ? RS / 4 : 0) << 16, x masks of the four planes
//   Y record (8 dwords): y01, y23, 4 (pad_slab v + y0 pitch), (any ? rows : 0) | y0 << 8, y masks of the four planes, 0, 0, 0
constexpr int RX_BIAS_X = 64, RX_BIAS_Y = 8;  // quads / rows

__global__ __launch_bounds__(256) void plan_rect_pack(SweepParams p, RectTables rt, int RS)
{
    const int NC = p.nchunks, dpad = rt.dpad;
    const int nx = p.tiles_x * p.V * NC * 4, ny = p.tiles_y * p.V * NC * 4;
    const bool live = (int)(blockIdx.x * blockDim.x + threadIdx.x) < nx + ny;
    int tid = min((int)(blockIdx.x * blockDim.x + threadIdx.x), nx + ny - 1);
    const int which = tid < nx ? 0 : 1;
    if (which == 1) tid -= nx;
    const int w = tid & 3, chunk = (tid >> 2) % NC, v = ((tid >> 2) / NC) % p.V, t = (tid >> 2) / (NC * p.V);
    const uint32_t *ent = (which == 0 ? rt.xt : rt.yt) + ((size_t)t * p.V + v) * dpad + chunk * RX_PC + w * RX_KW;
    const uint32_t box = (which == 0 ? rt.xbox : rt.ybox)[tid >> 2];
    const int org = (int)(box & 0xffffu), ext = (int)((box >> 16) & 0x7fffu);
    const bool any = (box >> 31) != 0u;
    const int size = which == 0 ? TILE_W : RX_TILE_H;
    uint32_t f[RX_KW], masks = 0u;
    int not_full = 0;
    for (int k = 0; k < RX_KW; k++) {
        const uint32_t e = ent[k];
        const int tex = ((int)(e & 0xfffffu) - RX_BIAS) >> 5;
        const int nout = (int)((e >> 20) & 127u), side = (int)((e >> 27) & 1u);
        const bool nothing = nout >= size || !any;
        // the in-frame pixels' texels lie inside the box (it is their bounding box); pixel 0's may lie up to `nout` texels before it
        const int bias = which == 0 ? RX_BIAS_X : RX_BIAS_Y;
        const int rel = tex - org + bias;
        const bool certified = (e & RX_UNIFORM) != 0u && (nothing || (rel >= 0 && rel < ext + 2 * bias));
        if (!certified) {
            f[k] = which == 0 ? 0x4000u : 0x8000u;
        } else {
            f[k] = nothing ? 0u : (uint32_t)rel * (which == 0 ? 4u : 4u * (uint32_t)RS);
            uint32_t m;
            if (which == 0)
                m = nothing ? 64u : (nout == 0 ? 0u : (uint32_t)nout | ((uint32_t)side << 7));
            else  // rows out of frame: the first `nout` (side 0) or the last `nout` (side 1)
                m = nothing ? 0xffu : (nout == 0 ? 0u : (side ? (0xffu << (RX_TILE_H - nout)) & 0xffu : (1u << nout) - 1u));
            masks |= m << (8 * k);
        }
        not_full += (!certified || ((masks >> (8 * k)) & 0xffu)) ? 1 : 0;
    }
    {
        const int n = wave_sum_i32(live ? not_full : 0);
        if ((threadIdx.x & 63) == 0 && n) atomicAdd(rt.stats + 2, n);
    }
    if (!live) return;
    if (which == 0) {
        uint32_t *rec = rt.xw + (size_t)tid * 8;
        const uint32_t *wv = rt.wt + (size_t)v * dpad + chunk * RX_PC + w * RX_KW;
        rec[0] = f[0] | (f[1] << 16);
        rec[1] = f[2] | (f[3] << 16);
        rec[2] = wv[0];
        rec[3] = wv[1];
        rec[4] = wv[2];
        rec[5] = wv[3];
        rec[6] = 4u * (uint32_t)org | ((any ? (uint32_t)RS / 4u : 0u) << 16);
        rec[7] = masks;
    } else {
        uint32_t *rec = rt.yr + (size_t)tid * 8;
        rec[0] = f[0] | (f[1] << 16);
        rec[1] = f[2] | (f[3] << 16);
        rec[2] = 4u * ((uint32_t)p.pad_slab * (uint32_t)(p.view_slot ? p.view_slot[v] : v) + (uint32_t)(org * p.pitch));  // (frame-store slot: mvs_sweep_handles)
        rec[3] = (any ? (uint32_t)ext : 0u) | ((uint32_t)org << 8);
        rec[4] = masks;
        rec[5] = rec[6] = rec[7] = 0u;
    }
}

// ------------------------------------------------------------------------------------------------------
// sweep kernel
// ------------------------------------------------------------------------------------------------------
// What the sweep kernel reads on every region comes in as kernel arguments (SGPRs); what only its rare paths, its prologue and its
// last lines need sits behind one pointer (RectCold, in device memory) and is fetched where it is used: the view loop is short of
// SGPRs, and a spilled SGPR costs vector instructions (v_readlane / v_writelane) in the loop.
struct RectCold {
    const uint8_t *main_img;
    const uint32_t *xt, *yt, *lut;
    const float *Q, *z;
    float *depth, *cost;
    int *index;
    float invW, invH;
    int dpad, pad_;
};

struct RectArgs {
    const uint32_t *__restrict__ quads;  // quad images of the side views, pad_slab dwords each
    const uint32_t *__restrict__ xw;
    const uint32_t *__restrict__ yr;
    uint32_t *__restrict__ volume;
    const RectCold *__restrict__ cold;
    uint2 *__restrict__ part;            // plane-split launches: partial bests [gridDim.y][P]
    size_t pad_slab;
    int pitch, W, H, D, V, v0, vcount, nchunks, chunk0, chunk1, cps, ty0, tyn, tiles_x;
    int slot_dw;   // dwords per LDS slot = 256 x copy instructions per region (one instruction fills 256 dwords); two slots
};

template <typename T>
__device__ __forceinline__ T cold_get(uintptr_t c, size_t off)
{
    return *(const __attribute__((address_space(4))) T *)(c + off);
}
#define RX_COLD(c, T, field) cold_get<T>((uintptr_t)(c), offsetof(RectCold, field))

// raw buffer resource over `bytes` bytes at p (gfx950: dword 3 = 0x00020000, 32-bit data format): buffer instructions take a
// wave-uniform resource + a 32-bit per-lane offset + a wave-uniform offset, so no access of the view loop needs a 64-bit per-lane
// address (through plain pointers the compiler forms one for every load, store and LDS copy: 2 VGPRs and 1-2 VALU instructions each).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000);
}

template <int RS, bool WRITE_VOLUME, bool FUSED>
__global__ __launch_bounds__(256, RX_WAVES_PER_SIMD) void sweep_fx_rect(RectArgs a)
{
    constexpr int UNITS = RS / 4;  // 16-byte units per region row
    constexpr int SLOT_BIAS_DW = RX_BIAS_X + RX_BIAS_Y * RS;  // dwords
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)smem;  // (through a generic pointer: a null check per use)

    const int band_tile = grouped_tile(blockIdx.x, a.tiles_x, a.tyn);
    if (band_tile < 0) return;
    const int tx = band_tile % a.tiles_x, ty = band_tile / a.tiles_x + a.ty0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = tx * TILE_W + lane;
    const int row0 = ty * RX_TILE_H;
    const bool col_ok = col < a.W;
    const int NC = a.nchunks;

    uint32_t Im255[8];
    {
        const uint8_t *img = RX_COLD(a.cold, const uint8_t *, main_img);
#pragma unroll
        for (int j = 0; j < 8; j++) Im255[j] = (col_ok && row0 + j < a.H) ? 255u * (uint32_t)img[(size_t)(row0 + j) * a.W + col] : 0u;
    }

    // per-lane source offsets (bytes) of this wavefront's copy instructions: instruction i = wave + 4 t fills LDS dwords
    // [256 i, 256 i + 256) of the slot = 16-byte units g = 64 i + lane of the dense [row][RS] region image
    uint32_t srcoff[RX_MAX_NI];
#pragma unroll
    for (int t = 0; t < RX_MAX_NI; t++) {
        const int g = (wave + 4 * t) * 64 + lane;
        srcoff[t] = 4u * (uint32_t)((g / UNITS) * a.pitch + (g % UNITS) * 4);
    }

    const int chunk_first = a.chunk0 + (int)blockIdx.y * a.cps;
    const int chunk_last = min(a.chunk1, chunk_first + a.cps);
    const int vend = a.v0 + a.vcount;
    const int nreg = (chunk_last - chunk_first) * a.vcount;

    // this wavefront's records of region (chunk, v), e = v NC + chunk: 8 dwords at xw_wg + 128 e bytes, 8 dwords at yr_wg + 128 e bytes;
    // both tables live in one allocation (a.xw < a.yr): one resource, two wave-uniform offsets
    const __amdgpu_buffer_rsrc_t rtab = make_rsrc(a.xw, 0xffffffffu);
    const uint32_t xw_wg = (uint32_t)(((size_t)tx * a.V * NC * 4 + wave) * 32);
    const uint32_t yr_wg = (uint32_t)((size_t)((const char *)a.yr - (const char *)a.xw) + ((size_t)ty * a.V * NC * 4 + wave) * 32);
    const __amdgpu_buffer_rsrc_t rquads = make_rsrc(a.quads, 0xffffffffu);

    // request a region into the slot at LDS dword `slot_dw0`: xsx = X record dword 6 (4 x0 | units per row << 16, 0 units if the box is
    // empty), ysrc / yn = Y record dwords 2 / 3 (byte offset of the box's first row in the quad images; rows | y0 << 8).  Rows of RS
    // quads, 64 16-byte units per instruction; only the last instruction of a region runs under a lane mask.
    auto issue_copy = [&](uint32_t xsx, uint32_t ysrc, uint32_t yn, uint32_t slot_dw0) {
        const int n = (int)((yn & 0xffu) * (xsx >> 16));
        const uint32_t src = (xsx & 0xffffu) + ysrc;
        uint32_t *dst = smem + slot_dw0 + SLOT_BIAS_DW + wave * 256;  // (the records' offsets are biased: pixel 0 of a partly visible tile lies before the box)
        const int left = n - wave * 64;  // units of the region this wavefront still has to copy (wave-uniform)
        auto copy = [&](int t, bool whole) {
            int lim = left - 256 * t;
            asm volatile("" : "+s"(lim));  // (or the compiler folds the uniform test that led here into this per-lane one: an EXEC mask and its branch per instruction)
            if (whole || lane < lim)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rquads, (__attribute__((address_space(3))) void *)(dst + t * 1024), 16, srcoff[t], src, 0, 0);
        };
        // whole instructions, then the region's last one under a lane mask; nothing left: no later instruction of this wavefront either
        static_assert(RX_MAX_NI == 3, "the nest below");
        if (left >= 64) {
            copy(0, true);
            if (left >= 320) {
                copy(1, true);
                if (left >= 576)
                    copy(2, true);
                else if (left > 512)
                    copy(2, false);
            } else if (left > 256) {
                copy(1, false);
            }
        } else if (left > 0) {
            copy(0, false);
        }
    };

    // The records of the coming regions travel through VECTOR registers (lane l holds dword l of the record; one v_readlane each when
    // the region's turn comes): as scalar loads they would have to stay in SGPRs across a whole region's sampling, and the compiler
    // spills them right after the load (a wait for the load, then v_writelane / v_readlane pairs) -- measured: 0.65 ms of loop skeleton.
    const uint32_t lane8 = 4u * (uint32_t)(lane & 7);
    auto load_x = [&](uint32_t xo) { return __builtin_amdgcn_raw_buffer_load_b32(rtab, lane8, xo, 0); };
    auto load_y = [&](uint32_t yo) { return __builtin_amdgcn_raw_buffer_load_b32(rtab, lane8, yo, 0); };
    auto rdl = [](uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); };
    // Region order of this workgroup: chunks outer, views inner.  A cursor = (views left in the chunk, byte offsets of the region's
    // X and Y records); past the workgroup's last region it stays there (prefetches re-read the last records).
    struct Cursor {
        int vleft, left;  // views left in this chunk after this one; regions left after this one
        uint32_t xo;      // byte offset of the region's X record; the Y records are walked in step: Y record at xo + ydelta
    };
    const uint32_t xstep = 128u * (uint32_t)NC;  // next view, same chunk
    const int vlast = a.vcount - 1;
    const uint32_t xwrap = 128u - xstep * (uint32_t)vlast;  // first view of the next chunk
    auto advance = [&](Cursor &c) {
        if (c.left <= 0) return;
        c.left--;
        const bool wrap = c.vleft == 0;
        c.xo += wrap ? xwrap : xstep;
        c.vleft = wrap ? vlast : c.vleft - 1;
    };
    const uint32_t x_first = xw_wg + 128u * (uint32_t)(a.v0 * NC + chunk_first);
    const uint32_t ydelta = yr_wg - xw_wg;

    uint32_t acc[8][RX_KW];
    uint32_t best[8];
    int bi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        best[j] = 0xffffffffu;  // plain comparison (chunk epilogue); the cross-multiplied one starts from (sum 1, count 0)
        bi[j] = -1;
#pragma unroll
        for (int k = 0; k < RX_KW; k++) acc[j][k] = 0u;
    }
    bool plain = true;    // depth selection: every cell so far carries the same count (see the chunk epilogue)
    uint32_t spacc = 0u;  // per plane of this wavefront (one byte each): views of the current chunk whose plane was NOT counted as a whole

    Cursor c2;  // region r + 2: the one whose records are fetched next
    c2.vleft = vlast;
    c2.left = nreg - 1;
    c2.xo = x_first;
    // prologue: the records of regions 0 and 1, the copy of region 0
    uint32_t x0r = load_x(c2.xo), y0r = load_y(c2.xo + ydelta);  // region r
    advance(c2);
    uint32_t x1r = load_x(c2.xo), y1r = load_y(c2.xo + ydelta);  // region r + 1
    advance(c2);
    uint32_t slot_cur = 0u, slot_nxt = (uint32_t)a.slot_dw;  // LDS dword offsets of the two slots
    if (nreg > 0) issue_copy(rdl(x0r, 6), rdl(y0r, 2), rdl(y0r, 3), slot_cur);
    int chunk = chunk_first, v = a.v0;

    for (int r = 0; r < nreg; r++) {
        // every copy and every record this wavefront asked for has landed ...
        wait_vm<0>();
        // ... and so have every other wavefront's; nobody reads region r - 1 any more
        __builtin_amdgcn_s_barrier();
        // request region r + 1 into the other slot and the records of region r + 2: in flight during this region's sampling
        if (r + 1 < nreg) issue_copy(rdl(x1r, 6), rdl(y1r, 2), rdl(y1r, 3), slot_nxt);
        const uint32_t x2r = load_x(c2.xo), y2r = load_y(c2.xo + ydelta);
        advance(c2);
        const uint32_t rsum = x0r + y0r;                            // (one vector add, two v_readlane: not four and two scalar adds)
        const uint32_t sum01 = rdl(rsum, 0), sum23 = rdl(rsum, 1);  // per plane: LDS byte offset, or a flag bit
        const uint32_t we[RX_KW] = {rdl(x0r, 2), rdl(x0r, 3), rdl(x0r, 4), rdl(x0r, 5)};
        const uint32_t fmask = 0xffffu;
        const uint32_t fld[RX_KW] = {sum01 & fmask, (sum01 >> 16) & fmask, sum23 & fmask, (sum23 >> 16) & fmask};

        // ---- sample region r ----
        const uint32_t special = (sum01 | sum23) & 0xc000c000u;
        {
            const uint32_t slot_byte = lds_base + slot_cur * 4u;
            uint32_t qd[1][8];
            // per plane: what is out of frame although the certificates hold (0: nothing -- with no flag in the field that is a FULL plane)
            const uint32_t xmasks = rdl(x0r, 7), ymasks = rdl(y0r, 4);
            const uint32_t anymask = xmasks | ymasks;
            if (__builtin_expect(anymask != 0u, 0)) {
#pragma unroll
                for (int k = 0; k < RX_KW; k++)
                    if (((anymask >> (8 * k)) & 0xffu) && !(fld[k] & 0xc000u)) spacc += 1u << (8 * k);  // this view's count does not go to every cell of the plane
            }
            // One asm statement per plane: LDS base in M0 (ds_read_addtid_b32: M0 + offset + 4 lane, no address register), weight word in
            // an SGPR, 8 reads, 8 v_dot4, 8 v_sad_u16 (all dot products before all differences: a v_sad right behind the v_dot4 it
            // consumes costs wait states).  A plane whose certificate failed (flag bits in the field) reads as well (somewhere in or past
            // the LDS: harmless) and runs its vector instructions with EXEC = 0; the block below does those planes.  The s_and between
            // the write of M0 and the first read is the wait state that pair needs.
            // A plane of a tile at the border of the side view (its byte of `am` is not 0) branches to the tail INSIDE the statement --
            // same registers, no second shape for the register allocator: EXEC = the lanes in frame (x byte: n | side << 7, n = 64:
            // none), then row by row EXEC = that or nothing (y byte: the rows that are out), each cell that gets the sample counting it
            // itself (+ 1 << 24).  The tail costs the planes that are FULL one taken s_branch.
#define RX_ROW_TAIL(j)                                                                                                   \
    "s_bitcmp0_b32 %[ym], %[yb" #j "]\n\ts_cselect_b64 exec, vcc, 0\n\tv_add_u32 %[a" #j "], 0x1000000, %[a" #j "]\n\t" \
    "v_sad_u16 %[a" #j "], %[q" #j "], %[i" #j "], %[a" #j "]\n\t"
#pragma unroll
            for (int k = 0; k < RX_KW; k++) {   // (experiment 512 -- every plane sampled twice: the work of 8 planes per wavefront, DESIGN A.5b -- has served and is gone: its
                                                 // run-time trip count no longer compiles with the immediates below)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
                asm volatile("s_add_u32 m0, %[slot], %[fld]\n\t"
                             "s_and_b32 vcc_lo, %[fld], 0xc000\n\t"
                             "ds_read_addtid_b32 %[q0] offset:%[o0]\n\tds_read_addtid_b32 %[q1] offset:%[o1]\n\tds_read_addtid_b32 %[q2] offset:%[o2]\n\tds_read_addtid_b32 %[q3] offset:%[o3]\n\t"
                             "ds_read_addtid_b32 %[q4] offset:%[o4]\n\tds_read_addtid_b32 %[q5] offset:%[o5]\n\tds_read_addtid_b32 %[q6] offset:%[o6]\n\tds_read_addtid_b32 %[q7] offset:%[o7]\n\t"
                             "s_cselect_b64 exec, 0, -1\n\t"
                             "s_bfe_u32 vcc_lo, %[am], %[kb]\n\t"  // this plane's byte of the masks: SCC = not 0
                             "s_cbranch_scc1 1f\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "v_dot4_u32_u8 %[q0], %[q0], %[w], 0\n\tv_dot4_u32_u8 %[q1], %[q1], %[w], 0\n\tv_dot4_u32_u8 %[q2], %[q2], %[w], 0\n\tv_dot4_u32_u8 %[q3], %[q3], %[w], 0\n\t"
                             "v_dot4_u32_u8 %[q4], %[q4], %[w], 0\n\tv_dot4_u32_u8 %[q5], %[q5], %[w], 0\n\tv_dot4_u32_u8 %[q6], %[q6], %[w], 0\n\tv_dot4_u32_u8 %[q7], %[q7], %[w], 0\n\t"
                             "v_sad_u16 %[a0], %[q0], %[i0], %[a0]\n\tv_sad_u16 %[a1], %[q1], %[i1], %[a1]\n\tv_sad_u16 %[a2], %[q2], %[i2], %[a2]\n\tv_sad_u16 %[a3], %[q3], %[i3], %[a3]\n\t"
                             "v_sad_u16 %[a4], %[q4], %[i4], %[a4]\n\tv_sad_u16 %[a5], %[q5], %[i5], %[a5]\n\tv_sad_u16 %[a6], %[q6], %[i6], %[a6]\n\tv_sad_u16 %[a7], %[q7], %[i7], %[a7]\n\t"
                             "2:\n\t"
                             "s_mov_b64 exec, -1\n\t"
                             "s_branch 9f\n\t"
                             // ---- the tail: lanes in frame from the x byte
                             "1:\n\t"
                             "s_bfe_u32 vcc_lo, %[xm], %[nb]\n\t"        // n (6 bits)
                             "s_bitcmp1_b32 %[xm], %[sb]\n\t"            // side
                             "s_cbranch_scc1 3f\n\t"
                             "s_lshl_b64 exec, exec, vcc_lo\n\t"         // the first n lanes are out
                             "s_branch 4f\n\t"
                             "3:\n\t"
                             "s_lshr_b64 exec, exec, vcc_lo\n\t"         // the last n lanes are out
                             "4:\n\t"
                             "s_bitcmp1_b32 %[xm], %[eb]\n\t"            // n = 64: nothing in frame
                             "s_cselect_b64 exec, 0, exec\n\t"
                             "s_mov_b64 vcc, exec\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "v_dot4_u32_u8 %[q0], %[q0], %[w], 0\n\tv_dot4_u32_u8 %[q1], %[q1], %[w], 0\n\tv_dot4_u32_u8 %[q2], %[q2], %[w], 0\n\tv_dot4_u32_u8 %[q3], %[q3], %[w], 0\n\t"
                             "v_dot4_u32_u8 %[q4], %[q4], %[w], 0\n\tv_dot4_u32_u8 %[q5], %[q5], %[w], 0\n\tv_dot4_u32_u8 %[q6], %[q6], %[w], 0\n\tv_dot4_u32_u8 %[q7], %[q7], %[w], 0\n\t"
                             RX_ROW_TAIL(0) RX_ROW_TAIL(1) RX_ROW_TAIL(2) RX_ROW_TAIL(3) RX_ROW_TAIL(4) RX_ROW_TAIL(5) RX_ROW_TAIL(6) RX_ROW_TAIL(7)
                             "s_branch 2b\n\t"
                             "9:"
                             : [q0] "=&v"(qd[0][0]), [q1] "=&v"(qd[0][1]), [q2] "=&v"(qd[0][2]), [q3] "=&v"(qd[0][3]), [q4] "=&v"(qd[0][4]), [q5] "=&v"(qd[0][5]), [q6] "=&v"(qd[0][6]), [q7] "=&v"(qd[0][7]),
                               [a0] "+v"(acc[0][k]), [a1] "+v"(acc[1][k]), [a2] "+v"(acc[2][k]), [a3] "+v"(acc[3][k]), [a4] "+v"(acc[4][k]), [a5] "+v"(acc[5][k]), [a6] "+v"(acc[6][k]), [a7] "+v"(acc[7][k])
                             : [slot] "s"(slot_byte), [fld] "s"(fld[k]), [w] "s"(we[k]), [am] "s"(anymask), [xm] "s"(xmasks), [ym] "s"(ymasks),
                               [i0] "v"(Im255[0]), [i1] "v"(Im255[1]), [i2] "v"(Im255[2]), [i3] "v"(Im255[3]), [i4] "v"(Im255[4]), [i5] "v"(Im255[5]), [i6] "v"(Im255[6]), [i7] "v"(Im255[7]),
                               [o0] "n"(0), [o1] "n"(RS * 4), [o2] "n"(RS * 8), [o3] "n"(RS * 12), [o4] "n"(RS * 16), [o5] "n"(RS * 20), [o6] "n"(RS * 24), [o7] "n"(RS * 28),
                               [kb] "n"(8 * k | (8 << 16)), [nb] "n"(8 * k | (6 << 16)), [sb] "n"(8 * k + 7), [eb] "n"(8 * k + 6),
                               [yb0] "n"(8 * k + 0), [yb1] "n"(8 * k + 1), [yb2] "n"(8 * k + 2), [yb3] "n"(8 * k + 3), [yb4] "n"(8 * k + 4), [yb5] "n"(8 * k + 5), [yb6] "n"(8 * k + 6), [yb7] "n"(8 * k + 7)
                             : "m0", "scc", "vcc");
#pragma clang diagnostic pop
            }
#undef RX_ROW_TAIL
            // the planes whose certificate failed (a rounding boundary inside the tile), or whose offsets did not fit the record
            if (special) {
                uintptr_t coldp = (uintptr_t)a.cold;
                asm volatile("" : "+s"(coldp));  // not loop-invariant for the optimiser: fetched here, not held in SGPRs over the loop
                const int dpad = RX_COLD(coldp, int, dpad);
                const cu32 xt = as_const<cu32>(RX_COLD(coldp, const uint32_t *, xt) + ((size_t)tx * a.V + v) * dpad + chunk * RX_PC + wave * RX_KW);
                const cu32 yt = as_const<cu32>(RX_COLD(coldp, const uint32_t *, yt) + ((size_t)ty * a.V + v) * dpad + chunk * RX_PC + wave * RX_KW);
                const uint32_t xsx = rdl(x0r, 6), yn = rdl(y0r, 3);
                const int x0 = (int)((xsx & 0xffffu) >> 2), y0 = (int)((yn >> 8) & 0x3fffu);
                const bool staged = (xsx >> 16) != 0u && (yn & 0xffu) != 0u;
#pragma unroll
                for (int k = 0; k < RX_KW; k++) {
                    if (!(fld[k] & 0xc000u)) continue;
                    spacc += 1u << (8 * k);  // not FULL: this view's count does not go to every cell of the plane
                    if (!staged) continue;
                    const uint32_t xe = xt[k], ye = yt[k];
                    const int nx = (int)((xe >> 20) & 127u), ny = (int)((ye >> 20) & 127u);
                    if ((xe & ye) & RX_UNIFORM) {
                        if (nx >= TILE_W || ny >= RX_TILE_H) continue;  // nothing in frame
                        // a lane range (from the X entry) and a row range (from the Y entry) are in frame: the fast path under a mask,
                        // the in-frame count per cell
                        const int tx0 = (int)(xe & 0xfffffu) - RX_BIAS, ty0 = (int)(ye & 0xfffffu) - RX_BIAS;
                        const uint32_t addr = 4u * (uint32_t)lane + slot_byte + 4u * (uint32_t)(SLOT_BIAS_DW + ((ty0 >> 5) - y0) * RS + ((tx0 >> 5) - x0));
                        const bool lane_in = ((xe >> 27) & 1u) ? lane < TILE_W - nx : lane >= nx;
                        const int jlo = ((ye >> 27) & 1u) ? 0 : ny, jhi = ((ye >> 27) & 1u) ? RX_TILE_H - ny : RX_TILE_H;
                        if (lane_in) {
#pragma unroll
                            for (int j = 0; j < 8; j++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(qd[0][j]) : "v"(addr), "n"(j * RS * 4));
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qd[0][0]), "+v"(qd[0][1]), "+v"(qd[0][2]), "+v"(qd[0][3]), "+v"(qd[0][4]), "+v"(qd[0][5]), "+v"(qd[0][6]), "+v"(qd[0][7]));
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                if (j >= jlo && j < jhi) acc[j][k] = sad_u16(__builtin_amdgcn_udot4(qd[0][j], we[k], 0u, false), Im255[j], acc[j][k] + (1u << 24));
                        }
                        continue;
                    }
                    // A certificate failed (a rounding boundary inside the tile: ~0.3 % of the planes on the SURVEY 8d ring): the contract's
                    // expression per lane (columns) and per row, then one pass per group of lanes that share a phase and a texel offset
                    // (usually two groups), each with wave-uniform weights and LDS bases like the fast path.
                    const cf32 q = as_const<cf32>(RX_COLD(coldp, const float *, Q) + 12 * v);
                    const float z = as_const<cf32>(RX_COLD(coldp, const float *, z))[min(chunk * RX_PC + wave * RX_KW + k, a.D - 1)];
                    const float invW = RX_COLD(coldp, float, invW), invH = RX_COLD(coldp, float, invH);
                    const cu32 lut = as_const<cu32>(RX_COLD(coldp, const uint32_t *, lut));
                    const uint32_t lo_bits = __builtin_bit_cast(uint32_t, RX_MAGIC + 132.0f);
                    const uint32_t hix_bits = lo_bits + 256u * (uint32_t)a.W, hiy_bits = lo_bits + 256u * (uint32_t)a.H;  // floats in [2^23, 2^24): ulp 1
                    const float r256 = rect_r256(q);
                    const uint32_t txb = __builtin_bit_cast(uint32_t, rect_tx(q, r256, z, col, invW));
                    const bool inx = col_ok && txb > lo_bits && txb < hix_bits;
                    const int tx3 = (int)((txb & 0x3fffffu) >> 3) - 32 * lane;
                    const uint32_t tyv = __builtin_bit_cast(uint32_t, rect_ty(q, r256, z, row0 + (lane & 7), invH));
                    unsigned long long remaining = __builtin_amdgcn_ballot_w64(inx);
                    while (remaining) {
                        const int t = __builtin_amdgcn_readlane(tx3, (int)__builtin_ctzll(remaining));
                        const bool mine = inx && tx3 == t;
                        remaining &= ~__builtin_amdgcn_ballot_w64(mine);
                        const uint32_t kx = (uint32_t)t & 31u;
                        const int ixrel = (t >> 5) - x0;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const uint32_t tyb = (uint32_t)__builtin_amdgcn_readlane((int)tyv, j);
                            if (tyb > lo_bits && tyb < hiy_bits && row0 + j < a.H) {
                                const uint32_t uy = tyb & 0x3fffffu;
                                const uint32_t w = lut[((uy >> 3) & 31u) * 32u + kx];
                                const uint32_t addr = 4u * (uint32_t)lane + slot_byte + 4u * (uint32_t)(SLOT_BIAS_DW + ((int)(uy >> 8) - y0) * RS + ixrel);
                                if (mine) {
                                    uint32_t quad;
                                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(quad) : "v"(addr));
                                    acc[j][k] = sad_u16(__builtin_amdgcn_udot4(quad, w, 0u, false), Im255[j], acc[j][k] + (1u << 24));
                                }
                            }
                        }
                    }
                }
            }
        }

        // ---- chunk epilogue ----
        if (v + 1 == vend) {
            const int d0 = chunk * RX_PC + wave * RX_KW;
            const size_t P = (size_t)a.W * a.H;
            const uint32_t pix0 = 4u * (uint32_t)(row0 * a.W + col);  // byte offset of this lane's first pixel inside a plane
            const int nrows = min(8, a.H - row0);
            // While every plane of every chunk so far was FULL for every view, every cell carries the same count and the packed cells
            // compare like their sums: "cell < best" (best starts at 0xffffffff) instead of the cross-multiplied comparison.  The
            // first chunk with a plane that is not FULL ends that for the rest of the workgroup (same packed cells, start value 1).
            if (FUSED && plain && spacc != 0u) {
                plain = false;
#pragma unroll
                for (int j = 0; j < 8; j++) best[j] = bi[j] < 0 ? 1u : best[j];
            }
            auto finish = [&](auto checked_rows, auto plain_compare) {
#pragma unroll
                for (int k = 0; k < RX_KW; k++) {
                    const uint32_t cntk = ((uint32_t)a.vcount - ((spacc >> (8 * k)) & 0xffu)) << 24;  // FULL planes: every cell gets the view's count
                    if (d0 + k < a.D) {
                        // one resource per plane (a volume can exceed the 4 GiB a resource spans), rows by the wave-uniform offset
                        const __amdgpu_buffer_rsrc_t rvol = make_rsrc(WRITE_VOLUME ? a.volume + (size_t)(d0 + k) * P : nullptr, 0xffffffffu);
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            if (!checked_rows.value || j < nrows) {
                                const uint32_t cell = acc[j][k] + cntk;
                                if (WRITE_VOLUME) __builtin_amdgcn_raw_buffer_store_b32(cell, rvol, pix0, 4u * (uint32_t)(j * a.W), 2);  // nt: written once, read by a later kernel
                                if (FUSED) {
                                    const bool better = plain_compare.value ? cell < best[j] : umul24u(cell & 0xffffffu, best[j] >> 24) < umul24u(best[j] & 0xffffffu, cell >> 24);
                                    best[j] = better ? cell : best[j];
                                    bi[j] = better ? d0 + k : bi[j];
                                }
                            }
                        }
                    }
                }
            };
            if (col_ok) {
                if (nrows == 8 && plain)
                    finish(std::false_type{}, std::true_type{});
                else if (nrows == 8)
                    finish(std::false_type{}, std::false_type{});
                else if (plain)
                    finish(std::true_type{}, std::true_type{});
                else
                    finish(std::true_type{}, std::false_type{});
            }
#pragma unroll
            for (int k = 0; k < RX_KW; k++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j][k] = 0u;
            spacc = 0u;
        }

        if (++v == vend) {
            v = a.v0;
            chunk++;
        }
        x0r = x1r;
        y0r = y1r;
        x1r = x2r;
        y1r = y2r;
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
        float *depth = RX_COLD(a.cold, float *, depth), *cost = RX_COLD(a.cold, float *, cost);
        int *index = RX_COLD(a.cold, int *, index);
        const float *zt = RX_COLD(a.cold, const float *, z);
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
                        const uint32_t lhs = umul24u(c.x & 0xffffffu, b >> 24), rhs = umul24u(b & 0xffffffu, c.x >> 24);
                        const bool take = lhs < rhs || (lhs == rhs && bidx >= 0 && (int)c.y < bidx);
                        b = take ? c.x : b;
                        bidx = take ? (int)c.y : bidx;
                    }
                }
                if (bidx < 0) b = 0u;  // no plane had a view in frame: the empty cell, as argmin_update_packed leaves it
                const size_t pix = (size_t)row * a.W + col;
                if (a.part) {
                    a.part[(size_t)blockIdx.y * P + pix] = make_uint2(b, (uint32_t)bidx);
                } else {  // store_best<CS_FIXED>
                    depth[pix] = bidx >= 0 ? zt[bidx] : MVS_BACKGROUND_DEPTH;
                    cost[pix] = bidx >= 0 ? cell_cost<CS_FIXED>(b & 0xffffffu, b >> 24) : __builtin_inff();
                    index[pix] = bidx;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------------
int ensure_fx_lut(mvs_ctx *ctx);  // sweep_fx.hip

bool rect_view_host(const float *q) { return q[1] == 0.0f && q[4] == 0.0f && q[8] == 0.0f && q[9] == 0.0f && q[10] == 0.0f && q[11] > 0.0f; }

struct RectSizes {
    size_t nx, ny, nw, nxw, nyr, nxb, nyb, total;
};

static RectSizes rect_sizes(const SweepParams &q)
{
    RectSizes z;
    const size_t dpad = (size_t)q.nchunks * RX_PC;
    z.nx = (size_t)q.tiles_x * q.V * dpad;
    z.ny = (size_t)q.tiles_y * q.V * dpad;
    z.nw = (size_t)q.V * dpad;
    z.nxb = (size_t)q.tiles_x * q.V * q.nchunks;
    z.nyb = (size_t)q.tiles_y * q.V * q.nchunks;
    z.nxw = z.nxb * 32;
    z.nyr = z.nyb * 32;
    z.total = 16 + z.nxw + z.nyr + 2 * z.nx + 2 * z.ny + z.nw + z.nxb + z.nyb + 64;  // + slack: prefetches read whole 8-dword lanes past a Y record
    return z;
}

static void rect_tables(mvs_ctx *ctx, const SweepParams &q, RectTables &rt)
{
    const RectSizes z = rect_sizes(q);
    uint32_t *base = (uint32_t *)ctx->rect_tab.ptr;
    rt.dpad = q.nchunks * RX_PC;
    rt.stats = (int *)base;  // 16 dwords
    rt.xw = base + 16;
    rt.yr = rt.xw + z.nxw;
    rt.xt = rt.yr + z.nyr;
    rt.yt = rt.xt + z.nx;
    rt.wt = rt.yt + z.ny;
    rt.xmm = rt.wt + z.nw;
    rt.ymm = rt.xmm + z.nx;
    rt.xbox = rt.ymm + z.ny;
    rt.ybox = rt.xbox + z.nxb;
}

static size_t rect_table_dwords(const SweepParams &q) { return rect_sizes(q).total; }

// Builds the tables for the current (views, planes) and decides whether the rectified kernel serves this plan: every view
// eligible (host check on the f32 view matrices) and every region box within a slot shape the kernel is compiled for (counters of
// pass B: one stream synchronisation per plan).  Called from sweep_fx_plan.
int sweep_rect_plan(mvs_ctx *ctx, PlanHook *between)
{
    struct RunHookOnce {  // whichever way this function leaves, the caller's work has been queued exactly once
        PlanHook *h;
        int rc = MVS_OK;
        bool done = false;
        int go()
        {
            if (!done && h) rc = h->run();
            done = true;
            return rc;
        }
    } hook{between};
    ctx->rect_ok = false;
    ctx->rect_cold_sent = false;  // the tables (and the cold block behind them) may move
    if (ctx->hooks.no_rect) return hook.go();
    if (ctx->V == 0) return hook.go();
    for (int v = 0; v < ctx->V; v++)
        if (!rect_view_host(ctx->q_host.data() + 12 * v)) return hook.go();  // a view that is not rectified: the general kernel
    // the kernel addresses the quad images through one buffer resource with 32-bit byte offsets
    if ((unsigned long long)ctx->pad_slab * (unsigned long long)(ctx->views_in_store ? ctx->store_cap : ctx->V) * 4ull >= (1ull << 32)) return hook.go();
    SweepParams q;
    fill_params(ctx, q, 0, ctx->V, RX_TILE_H, RX_PC);
    int rc;
    if ((rc = ensure(ctx, ctx->rect_tab, rect_table_dwords(q) * sizeof(uint32_t) + sizeof(RectCold) + 256)) || (rc = ensure_fx_lut(ctx))) {
        (void)hook.go();
        return rc;
    }
    if (!ctx->plan_event) MVS_HIP(ctx, hipEventCreateWithFlags(&ctx->plan_event, hipEventDisableTiming));
    RectTables rt;
    rect_tables(ctx, q, rt);
    MVS_HIP(ctx, hipMemsetAsync(rt.stats, 0, 64, ctx->stream));
    const size_t na = (size_t)(q.tiles_x + q.tiles_y + 1) * q.V * rt.dpad;
    plan_rect_axis<<<(unsigned)((na + 255) / 256), 256, 0, ctx->stream>>>(q, rt, (const uint32_t *)ctx->fx_lut.ptr);
    const size_t nb = (size_t)(q.tiles_x + q.tiles_y) * q.V * q.nchunks;
    plan_rect_box<<<(unsigned)((nb + 255) / 256), 256, 0, ctx->stream>>>(q, rt);
    MVS_HIP(ctx, hipGetLastError());
    int stats[4] = {0, 0, 0, 0};
    MVS_HIP(ctx, hipMemcpyAsync(stats, rt.stats, sizeof(stats), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipEventRecord(ctx->plan_event, ctx->stream));
    // the caller's work (quad images of the new views) goes into the queue NOW: it runs while the host waits for the counters
    if ((rc = hook.go())) return rc;
    MVS_HIP(ctx, hipEventSynchronize(ctx->plan_event));
    const int max_rw = stats[0], max_rh = stats[1];
    int rs = 0;
    if (ctx->hooks.rect_verbose) fprintf(stderr, "sweep_rect_plan: widest box %d quads, tallest %d rows\n", max_rw, max_rh);
    for (int cand : {64, 84, 96, 128})  // row strides the kernel is compiled for
        if (max_rw <= cand) {
            rs = cand;
            break;
        }
    if (!rs || max_rw <= 0 || max_rh <= 0 || max_rh > 32) return MVS_OK;  // wide baselines / few planes: boxes too large for the slots
    const int units = rs / 4;
    const int instrs = div_up(max_rh * units, 64);  // 1 KiB copy instructions per region
    if (instrs > 4 * RX_MAX_NI || instrs * 1024 >= 16384) return MVS_OK;
    if (4 * (rs + RX_BIAS_X) + 4 * rs * (max_rh + RX_BIAS_Y) >= 16384) return MVS_OK;  // the records hold 14-bit (biased) LDS offsets
    ctx->rect_rs = rs;
    ctx->rect_slot_dw = instrs * 256;
    ctx->rect_dpad = rt.dpad;
    plan_rect_pack<<<(unsigned)((4 * nb + 255) / 256), 256, 0, ctx->stream>>>(q, rt, rs);
    MVS_HIP(ctx, hipGetLastError());
    // the cold block (device memory, after the tables)
    RectCold cold;
    cold.main_img = main_image_ptr(ctx);
    cold.xt = rt.xt;
    cold.yt = rt.yt;
    cold.lut = (const uint32_t *)ctx->fx_lut.ptr;
    cold.Q = (const float *)ctx->qmats.ptr;
    cold.z = (const float *)ctx->ztab.ptr;
    cold.depth = (float *)ctx->depth.ptr;
    cold.cost = (float *)ctx->cost.ptr;
    cold.index = (int *)ctx->index.ptr;
    cold.invW = q.invW;
    cold.invH = q.invH;
    cold.dpad = rt.dpad;
    cold.pad_ = 0;
    ctx->rect_cold_host.resize(sizeof(RectCold));
    memcpy(ctx->rect_cold_host.data(), &cold, sizeof(RectCold));
    ctx->rect_ok = true;
    return MVS_OK;
}

template <int RS>
static int launch_rect(mvs_ctx *ctx, const RectArgs &a, dim3 grid, size_t lds, bool vol, bool fused)
{
    auto go = [&](auto kernel) -> int {
        MVS_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kernel<<<grid, 256, lds, ctx->stream>>>(a);
        MVS_HIP(ctx, hipGetLastError());
        return MVS_OK;
    };
    if (vol && fused) return go(sweep_fx_rect<RS, true, true>);
    if (vol) return go(sweep_fx_rect<RS, true, false>);
    return go(sweep_fx_rect<RS, false, true>);
}

// launch of the rectified sweep; `p` carries the plane / row / view ranges.  Returns the split count like sweep_fx_launch.
int sweep_rect_launch(mvs_ctx *ctx, SweepParams &p, bool vol, bool fused, unsigned flags)
{
    RectTables rt;
    rect_tables(ctx, p, rt);
    // the cold block follows the tables; its output pointers can change between plan and launch (ensure_outputs): re-sent when they do
    RectCold *cold_dev = (RectCold *)((uint32_t *)ctx->rect_tab.ptr + rect_table_dwords(p));
    RectCold cold;
    memcpy(&cold, ctx->rect_cold_host.data(), sizeof(RectCold));
    cold.main_img = main_image_ptr(ctx);
    cold.depth = (float *)ctx->depth.ptr;
    cold.cost = (float *)ctx->cost.ptr;
    cold.index = (int *)ctx->index.ptr;
    if (!ctx->rect_cold_sent || memcmp(&cold, ctx->rect_cold_host.data(), sizeof(RectCold)) != 0) {
        memcpy(ctx->rect_cold_host.data(), &cold, sizeof(RectCold));
        MVS_HIP(ctx, hipMemcpyAsync(cold_dev, ctx->rect_cold_host.data(), sizeof(RectCold), hipMemcpyHostToDevice, ctx->stream));
        ctx->rect_cold_sent = true;
    }
    RectArgs a;
    a.quads = p.quads;
    a.xw = rt.xw;
    a.yr = rt.yr;
    a.volume = p.volume;
    a.cold = cold_dev;
    a.part = nullptr;
    a.pad_slab = p.pad_slab;
    a.pitch = p.pitch;
    a.W = p.W;
    a.H = p.H;
    a.D = p.D;
    a.V = p.V;
    a.v0 = p.v0;
    a.vcount = p.vcount;
    a.nchunks = p.nchunks;
    a.chunk0 = p.chunk0;
    a.chunk1 = p.chunk1;
    a.ty0 = p.ty0;
    a.tyn = p.tyn;
    a.tiles_x = p.tiles_x;
    a.slot_dw = ctx->rect_slot_dw;
    size_t lds = ((size_t)2 * a.slot_dw + RX_BIAS_X + (size_t)RX_BIAS_Y * ctx->rect_rs) * 4;  // two slots, the second one's data ends a bias further on
    if (fused) lds = lds < 16384 ? 16384 : lds;  // the cross-wavefront depth selection borrows 16 KiB
    if (lds > 160 * 1024) return fail(ctx, MVS_EINVAL, "sweep_rect_launch: %zu bytes of LDS", lds);

    const int groups = div_up(p.tiles_x, 2) * div_up(p.tyn, 4);
    const int nch = p.chunk1 - p.chunk0, tiles = p.tiles_x * p.tyn;
    int want = (int)((flags >> 16) & 0xffu);
    if (!want) want = div_up(16 * ctx->num_cus, tiles);
    p.cps = div_up(nch, max(1, min(want, nch)));
    a.cps = p.cps;
    const int nsplit = div_up(nch, p.cps);
    int rc;
    if (fused && nsplit > 1) {
        if ((rc = ensure(ctx, ctx->best_parts, (size_t)nsplit * ctx->W * ctx->H * sizeof(uint2)))) return rc;
        p.part = (uint2 *)ctx->best_parts.ptr;
        a.part = p.part;
    }
    const dim3 grid((unsigned)(div_up(groups, 8) * 64), (unsigned)nsplit);
    rc = ctx->rect_rs == 64 ? launch_rect<64>(ctx, a, grid, lds, vol, fused) : ctx->rect_rs == 84 ? launch_rect<84>(ctx, a, grid, lds, vol, fused)
       : ctx->rect_rs == 96 ? launch_rect<96>(ctx, a, grid, lds, vol, fused) : launch_rect<128>(ctx, a, grid, lds, vol, fused);
    if (rc) return rc;
    return nsplit;
}

}  // namespace mvs
