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
// Region pipeline.  Per (chunk, view) the planner's box of the view's quad image is copied into one of S LDS slots by
// global_load_lds_dwordx4 (dense rows of RS quads, 1 KiB = 64 lanes x 16 B per instruction, per-lane source offsets fixed for the
// launch).  Copies run L = S - 1 regions ahead; a wavefront waits for ITS OWN copies of the region at hand with a counted
// s_waitcnt vmcnt(n) -- the younger regions' copies stay in flight -- and one s_barrier per region makes every wavefront's part
// visible and frees the slot the next copy goes to.  Descriptors and the X / Y / W table entries come in by scalar loads, one
// region ahead.  LDS reads in the sample loop are inline asm (the compiler would order every ds_read behind ALL pending LDS-DMA).
#include "sweep_shared.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace mvs {

namespace {

constexpr float RX_MAGIC = 12582912.0f;  // 1.5 * 2^23 (FX_MAGIC of sweep_fx.hip)
constexpr int RX_TILE_H = 8, RX_PC = 16, RX_KW = 4;  // tile rows, planes per chunk, planes per wavefront
constexpr int RX_MAX_NI = 3;                          // copy instructions per wavefront and region, at most
constexpr int RX_WAVES_PER_SIMD = 4;                  // launch bound: <= 128 VGPRs

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
// planner: X / Y / W tables
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

// One thread per (tile column or tile row, view, plane), evaluating every pixel of the tile.  Entry:
//   bits  0-19  t0 + RX_BIAS, t0 = (u >> 3) - 32 i of the in-frame pixels (i = index in the tile, u = 1/256-texel coordinate):
//               phase in the low 5 bits, integer texel of the tile's pixel 0 (extrapolated, possibly left of the image) above
//   bits 20-26  n = pixels of the tile that are out of frame (all of the tile's pixels: 64 resp. 8)
//   bit  27     which end: 0 = the first n pixels are out, 1 = the last n
//   bit  30     certificate: the out-of-frame pixels are exactly those n, every other pixel has (u >> 3) - 32 i == t0, and
//               the phase is the view's nominal one for this plane (the phase at the image centre: the W table's)
// Pixels past the image edge (ragged last tile) count as matching.
__global__ __launch_bounds__(256) void plan_rect_axis(SweepParams p, int dpad, uint32_t *__restrict__ xt, uint32_t *__restrict__ yt, uint32_t *__restrict__ wt,
                                                      const uint32_t *__restrict__ lut)
{
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
        wt[tid] = lut[nomy * 32u + nomx];
        return;
    }
    const int size = which == 0 ? TILE_W : RX_TILE_H;
    const int first = t * size;
    const int count = min(first + size, which == 0 ? p.W : p.H) - first;  // pixels of the tile inside the image
    int nin = 0, first_in = -1, last_in = -1;
    int t0 = 0;
    bool uniform = elig;
    for (int i = 0; i < count; i++) {
        const float T = which == 0 ? rect_tx(q, r256, z, first + i, p.invW) : rect_ty(q, r256, z, first + i, p.invH);
        if (!(T > RX_MAGIC + 132.0f && T < (which == 0 ? hix : hiy))) continue;
        const int ti = (int)((__builtin_bit_cast(uint32_t, T) & 0x3fffffu) >> 3) - 32 * i;
        if (nin == 0) {
            t0 = ti;
            first_in = i;
        }
        uniform = uniform && ti == t0;
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
    (which == 0 ? xt : yt)[tid] = ((uint32_t)(t0 + RX_BIAS) & 0xfffffu) | ((uint32_t)nout << 20) | ((uint32_t)side << 27) | (uniform ? RX_UNIFORM : 0u);
}

// ------------------------------------------------------------------------------------------------------
// sweep kernel
// ------------------------------------------------------------------------------------------------------
struct RectParams {
    const uint32_t *__restrict__ xt;   // [tiles_x][V][dpad]
    const uint32_t *__restrict__ yt;   // [tiles_y][V][dpad]
    const uint32_t *__restrict__ wt;   // [V][dpad]
    const uint32_t *__restrict__ lut;  // 32 x 32 weight table (slow path)
    int dpad;
    int slot_dw;   // dwords per LDS slot = 256 x instrs (one copy instruction fills 256 dwords); after the S slots: 256 dwords nobody reads
    int nslots;    // S
    int instrs;    // copy instructions per region (all four wavefronts together)
    int ni;        // copy instructions per wavefront and region = ceil(instrs / 4), the ones past `instrs` going to the unread dwords
};

// Everything a plane's fast paths need, decoded from one X and one Y entry (wave-uniform)
enum PlaneKind : int { PK_NONE = 0, PK_FULL = 1, PK_MASKED = 2, PK_SEMI = 3 };

template <int RS, bool WRITE_VOLUME, bool FUSED>
__global__ __launch_bounds__(256, RX_WAVES_PER_SIMD) void sweep_fx_rect(SweepParams p, RectParams rp)
{
    constexpr int UNITS = RS / 4;  // 16-byte units per region row
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)smem;

    const int band_tile = grouped_tile(blockIdx.x, p.tiles_x, p.tyn);
    if (band_tile < 0) return;
    const int tx = band_tile % p.tiles_x, ty = band_tile / p.tiles_x + p.ty0;
    const int tile = ty * p.tiles_x + tx;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = tx * TILE_W + lane;
    const int row0 = ty * RX_TILE_H;
    const bool col_ok = col < p.W;
    const size_t P = (size_t)p.W * p.H;

    uint32_t Im255[8];
#pragma unroll
    for (int j = 0; j < 8; j++) Im255[j] = (col_ok && row0 + j < p.H) ? 255u * (uint32_t)p.main_img[(size_t)(row0 + j) * p.W + col] : 0u;

    // per-lane source offsets (dwords) of this wavefront's copy instructions: instruction i = wave + 4 t fills LDS dwords
    // [256 i, 256 i + 256) of the slot = 16-byte units g = 64 i + lane of the dense [row][RS] region image
    uint32_t srcoff[RX_MAX_NI];
#pragma unroll
    for (int t = 0; t < RX_MAX_NI; t++) {
        const int g = (wave + 4 * t) * 64 + lane;
        srcoff[t] = (uint32_t)((g / UNITS) * p.pitch + (g % UNITS) * 4);
    }

    const int chunk_first = p.chunk0 + (int)blockIdx.y * p.cps;
    const int chunk_last = min(p.chunk1, chunk_first + p.cps);
    const int nreg = (chunk_last - chunk_first) * p.vcount;
    const int S = rp.nslots, L = S - 1;
    const int ni = rp.ni;

    // request region (chunk, v) into `slot`: always exactly `ni` copy instructions per wavefront, so that a counted vmcnt wait
    // can name a region; an instruction with no unit of the region to copy (short regions, SKIP mode) copies one
    // harmless unit from the head of the view's image with lane 0 (into slot space the region does not use, or past the slots)
    auto issue_copy = [&](int chunk, int v, int slot) {
        const u32x2 d = as_const<cu2>(p.plan)[((size_t)tile * p.nchunks + chunk) * p.V + v];
        const unsigned mode = (d.y >> 16) & 7u;
        const bool staged = mode == RX_FAST || mode == RX_BORDER;
        const int x0 = min((int)(d.x & 0xffffu), p.pitch - RS), y0 = (int)(d.x >> 16), rh = (int)((d.y >> 8) & 0xffu);
        const int n = staged ? rh * UNITS : 0;
        const uint32_t *src = p.quads + p.pad_slab * v + (staged ? (uint32_t)(y0 * p.pitch + x0) : 0u);
#pragma unroll
        for (int t = 0; t < RX_MAX_NI; t++) {
            if (t < ni) {
                const int i = wave + 4 * t;
                const int left = n - i * 64;  // units of the region this instruction still has to copy (wave-uniform)
                const bool act = lane < left;
                const bool dummy = left <= 0 && lane == 0;
                uint32_t *dst = smem + (i < rp.instrs ? slot * rp.slot_dw + i * 256 : S * rp.slot_dw);
                if (act || dummy)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (act ? srcoff[t] : 0u)),
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            }
        }
    };

    // the table entries and the descriptor of region (chunk, v) for this wavefront's four planes
    struct RegionInfo {
        u32x4 x, y, w;
        u32x2 d;
    };
    auto load_info = [&](int chunk, int v) {
        RegionInfo r;
        const int dk = chunk * RX_PC + wave * RX_KW;
        r.x = *as_const<cu4>(rp.xt + ((size_t)tx * p.V + v) * rp.dpad + dk);
        r.y = *as_const<cu4>(rp.yt + ((size_t)ty * p.V + v) * rp.dpad + dk);
        r.w = *as_const<cu4>(rp.wt + (size_t)v * rp.dpad + dk);
        r.d = as_const<cu2>(p.plan)[((size_t)tile * p.nchunks + chunk) * p.V + v];
        return r;
    };

    uint32_t acc[8][RX_KW];
    uint32_t best[8];
    int bi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        best[j] = 1u;  // (sum 1, count 0): "s * bc < bs * c" holds for the first cell with a view in frame, never for an empty one
        bi[j] = -1;
#pragma unroll
        for (int k = 0; k < RX_KW; k++) acc[j][k] = 0u;
    }
    uint32_t cnt[RX_KW] = {0u, 0u, 0u, 0u};  // views counted once for every cell of a plane (wave-uniform)

    // prologue: the first L regions
    {
        int c = chunk_first, v = p.v0;
        for (int i = 0; i < L && i < nreg; i++) {
            issue_copy(c, v, i);
            if (++v == p.v0 + p.vcount) {
                v = p.v0;
                c++;
            }
        }
    }
    int ca = chunk_first, va = p.v0;  // region r + L (the next one to request)
    for (int i = 0; i < L; i++)
        if (++va == p.v0 + p.vcount) {
            va = p.v0;
            ca++;
        }
    int slot_a = L % S, slot_c = 0;
    int chunk = chunk_first, v = p.v0;
    RegionInfo cur = nreg > 0 ? load_info(chunk, v) : RegionInfo{};
    const uint32_t lane4 = lds_base + 4u * (uint32_t)lane;
    const uint32_t lo_bits = __builtin_bit_cast(uint32_t, RX_MAGIC + 132.0f);
    const uint32_t hix_bits = lo_bits + 256u * (uint32_t)p.W, hiy_bits = lo_bits + 256u * (uint32_t)p.H;  // floats in [2^23, 2^24): ulp 1

    for (int r = 0; r < nreg; r++) {
        // next region's table entries: in flight during this region's sampling
        int cn = chunk, vn = v + 1;
        if (vn == p.v0 + p.vcount) {
            vn = p.v0;
            cn++;
        }
        RegionInfo nxt = cur;
        if (r + 1 < nreg) nxt = load_info(cn, vn);

        // this wavefront's copies of region r have landed (the copies of regions r + 1 .. r + L - 1 stay in flight) ...
        wait_vmcnt(r + L - 1 < nreg ? (L - 1) * ni : 0);
        // ... and so have every other wavefront's; nobody reads region r - 1 any more
        __builtin_amdgcn_s_barrier();
        if (r + L < nreg) {
            issue_copy(ca, va, slot_a);
            if (++va == p.v0 + p.vcount) {
                va = p.v0;
                ca++;
            }
            if (++slot_a == S) slot_a = 0;
        }

        // ---- sample region r ----
        const unsigned mode = (cur.d.y >> 16) & 7u;
        if (mode == RX_FAST || mode == RX_BORDER) {
            const int x0 = min((int)(cur.d.x & 0xffffu), p.pitch - RS), y0 = (int)(cur.d.x >> 16);
            const uint32_t xe[RX_KW] = {cur.x.x, cur.x.y, cur.x.z, cur.x.w}, ye[RX_KW] = {cur.y.x, cur.y.y, cur.y.z, cur.y.w};
            const uint32_t we[RX_KW] = {cur.w.x, cur.w.y, cur.w.z, cur.w.w};
            const uint32_t slot_byte = (uint32_t)(slot_c * rp.slot_dw) * 4u;
            uint32_t qd[2][8];
            int kind[RX_KW];
#pragma unroll
            for (int k = 0; k < RX_KW; k++) {
                const int nx = (int)((xe[k] >> 20) & 127u), ny = (int)((ye[k] >> 20) & 127u);
                kind[k] = !((xe[k] & ye[k]) & RX_UNIFORM) ? PK_SEMI : (nx >= TILE_W || ny >= RX_TILE_H) ? PK_NONE : (nx | ny) ? PK_MASKED : PK_FULL;
            }
#pragma unroll
            for (int k = 0; k <= RX_KW; k++) {
                if (k < RX_KW && (kind[k] == PK_FULL || kind[k] == PK_MASKED)) {
                    const int tx0 = (int)(xe[k] & 0xfffffu) - RX_BIAS, ty0 = (int)(ye[k] & 0xfffffu) - RX_BIAS;
                    const uint32_t addr = lane4 + slot_byte + 4u * (uint32_t)(((ty0 >> 5) - y0) * RS + ((tx0 >> 5) - x0));
                    if (kind[k] == PK_FULL) cnt[k] += 1u << 24;
#pragma unroll
                    for (int j = 0; j < 8; j++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(qd[k & 1][j]) : "v"(addr), "n"(j * RS * 4));
                }
                if (k > 0 && (kind[k - 1] == PK_FULL || kind[k - 1] == PK_MASKED)) {
                    const int b = (k - 1) & 1;
                    // LDS reads return in order: at most the 8 reads of plane k outstanding <=> plane k - 1's have landed.  The loaded
                    // registers are operands of the wait, so their consumers cannot be scheduled above it.
                    if (k < RX_KW && (kind[k] == PK_FULL || kind[k] == PK_MASKED))
                        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(qd[b][0]), "+v"(qd[b][1]), "+v"(qd[b][2]), "+v"(qd[b][3]), "+v"(qd[b][4]), "+v"(qd[b][5]), "+v"(qd[b][6]), "+v"(qd[b][7]));
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qd[b][0]), "+v"(qd[b][1]), "+v"(qd[b][2]), "+v"(qd[b][3]), "+v"(qd[b][4]), "+v"(qd[b][5]), "+v"(qd[b][6]), "+v"(qd[b][7]));
                    if (kind[k - 1] == PK_FULL) {
#pragma unroll
                        for (int j = 0; j < 8; j++) acc[j][k - 1] = sad_u16(__builtin_amdgcn_udot4(qd[b][j], we[k - 1], 0u, false), Im255[j], acc[j][k - 1]);
                    } else {
                        // part of the tile is out of frame in this plane: a lane range (from the X entry) and a row range (from the Y entry)
                        const int nx = (int)((xe[k - 1] >> 20) & 127u), ny = (int)((ye[k - 1] >> 20) & 127u);
                        const bool lane_in = ((xe[k - 1] >> 27) & 1u) ? lane < TILE_W - nx : lane >= nx;
                        const int jlo = ((ye[k - 1] >> 27) & 1u) ? 0 : ny, jhi = ((ye[k - 1] >> 27) & 1u) ? RX_TILE_H - ny : RX_TILE_H;
                        if (lane_in) {
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                if (j >= jlo && j < jhi)
                                    acc[j][k - 1] = sad_u16(__builtin_amdgcn_udot4(qd[b][j], we[k - 1], 0u, false), Im255[j], acc[j][k - 1] + (1u << 24));
                        }
                    }
                }
                if (k < RX_KW && kind[k] == PK_SEMI) {
                    // A certificate failed (a rounding boundary inside the tile, ~0.3 % of the planes; or a view that is not rectified):
                    // the contract's expression per lane (columns) and per row, then one pass per group of lanes that share a phase and
                    // a texel offset (usually two groups), each with wave-uniform weights and LDS bases like the fast path.
                    const cf32 q = as_const<cf32>(p.Q + 12 * v);
                    const float z = as_const<cf32>(p.z)[min(chunk * RX_PC + wave * RX_KW + k, p.D - 1)];
                    const float r256 = rect_r256(q);
                    const float Tx = rect_tx(q, r256, z, col, p.invW);
                    const uint32_t txb = __builtin_bit_cast(uint32_t, Tx);
                    const bool inx = col_ok && txb > lo_bits && txb < hix_bits;
                    const int tx3 = (int)((txb & 0x3fffffu) >> 3) - 32 * lane;
                    const uint32_t tyv = __builtin_bit_cast(uint32_t, rect_ty(q, r256, z, row0 + (lane & 7), p.invH));
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
                            if (tyb > lo_bits && tyb < hiy_bits && row0 + j < p.H) {
                                const uint32_t uy = tyb & 0x3fffffu;
                                const uint32_t w = as_const<cu32>(rp.lut)[((uy >> 3) & 31u) * 32u + kx];
                                const uint32_t addr = lane4 + slot_byte + 4u * (uint32_t)(((int)(uy >> 8) - y0) * RS + ixrel);
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
        if (v + 1 == p.v0 + p.vcount) {
            const int d0 = chunk * RX_PC + wave * RX_KW;
#pragma unroll
            for (int k = 0; k < RX_KW; k++) {
                if (d0 + k < p.D) {
                    uint32_t *const vol_plane = WRITE_VOLUME ? p.volume + (size_t)(d0 + k) * P : nullptr;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        if (col_ok && row0 + j < p.H) {
                            const uint32_t cell = acc[j][k] + cnt[k];
                            if (WRITE_VOLUME) __builtin_nontemporal_store(cell, vol_plane + (uint32_t)((row0 + j) * p.W + col));
                            if (FUSED) {
                                const bool better = umul24u(cell & 0xffffffu, best[j] >> 24) < umul24u(best[j] & 0xffffffu, cell >> 24);
                                best[j] = better ? cell : best[j];
                                bi[j] = better ? d0 + k : bi[j];
                            }
                        }
                    }
                }
                cnt[k] = 0u;
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j][k] = 0u;
            }
        }

        cur = nxt;
        chunk = cn;
        v = vn;
        if (++slot_c == S) slot_c = 0;
    }

    // ---- depth selection across the four wavefronts (each holds the best of its own planes): lowest cost, ties -> lowest plane ----
    if (FUSED) {
        __syncthreads();  // every copy has landed and every sample loop is done: the slots are free
        uint2 *ex = (uint2 *)smem;  // [wave][row][lane]
#pragma unroll
        for (int j = 0; j < 8; j++) ex[(wave * 8 + j) * 64 + lane] = make_uint2(best[j], (uint32_t)bi[j]);
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const int j = wave * 2 + jj;
            const int row = row0 + j;
            if (col_ok && row < p.H) {
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
                const size_t pix = (size_t)row * p.W + col;
                if (p.part)
                    p.part[(size_t)blockIdx.y * P + pix] = make_uint2(b, (uint32_t)bidx);
                else
                    store_best<CS_FIXED>(p, pix, b & 0xffffffu, b >> 24, bidx);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------------
int ensure_fx_lut(mvs_ctx *ctx);  // sweep_fx.hip

bool rect_view_host(const float *q) { return q[1] == 0.0f && q[4] == 0.0f && q[8] == 0.0f && q[9] == 0.0f && q[10] == 0.0f && q[11] > 0.0f; }

// Builds the X / Y / W tables for the current (views, planes) and decides whether the rectified kernel serves this plan: every
// view eligible and every staged region of the fixed-sampler plan within a slot shape the kernel is compiled for.
// Called after plan_regions_fx (whose counters it reads back: one stream synchronisation per plan).
int sweep_rect_plan(mvs_ctx *ctx)
{
    ctx->rect_ok = false;
    if (getenv("MVS_NO_RECT")) return MVS_OK;
    int elig = 0;
    for (int v = 0; v < ctx->V; v++) elig += rect_view_host(ctx->q_host.data() + 12 * v) ? 1 : 0;
    if (ctx->V == 0 || elig < ctx->V) return MVS_OK;  // a view that is not rectified would take the kernel's slowest path for all its planes
    int stats[4] = {0, 0, 0, 0};
    MVS_HIP(ctx, hipMemcpyAsync(stats, ctx->plan_stats.ptr, sizeof(stats), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int max_rw = stats[2], max_rh = stats[3];
    if (stats[0] > 0) return MVS_OK;  // regions beyond the LDS image of the general kernel: wide baselines, not this kernel's case
    int rs = 0;
    for (int cand : {96, 128})
        if (max_rw <= cand && ctx->pad_pitch >= cand) {
            rs = cand;
            break;
        }
    if (!rs || max_rh <= 0 || max_rh > 32) return MVS_OK;
    const int units = rs / 4;
    const int instrs = div_up(max_rh * units, 64);         // 1 KiB copy instructions per region
    const int ni = div_up(instrs, 4);
    if (ni > RX_MAX_NI) return MVS_OK;
    ctx->rect_rs = rs;
    ctx->rect_ni = ni;
    ctx->rect_instrs = instrs;
    ctx->rect_slot_dw = instrs * 256;

    SweepParams q;
    fill_params(ctx, q, 0, ctx->V, RX_TILE_H, RX_PC);
    const int dpad = q.nchunks * RX_PC;
    const size_t nx = (size_t)q.tiles_x * q.V * dpad, ny = (size_t)q.tiles_y * q.V * dpad, nw = (size_t)q.V * dpad;
    int rc;
    if ((rc = ensure(ctx, ctx->rect_tab, (nx + ny + nw) * sizeof(uint32_t) + 64))) return rc;
    if ((rc = ensure_fx_lut(ctx))) return rc;
    uint32_t *xt = (uint32_t *)ctx->rect_tab.ptr, *yt = xt + nx, *wt = yt + ny;
    plan_rect_axis<<<(unsigned)((nx + ny + nw + 255) / 256), 256, 0, ctx->stream>>>(q, dpad, xt, yt, wt, (const uint32_t *)ctx->fx_lut.ptr);
    MVS_HIP(ctx, hipGetLastError());
    ctx->rect_dpad = dpad;
    ctx->rect_ok = true;
    return MVS_OK;
}

template <int RS>
static int launch_rect(mvs_ctx *ctx, const SweepParams &p, const RectParams &rp, dim3 grid, size_t lds, bool vol, bool fused)
{
    auto go = [&](auto kernel) -> int {
        MVS_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kernel<<<grid, 256, lds, ctx->stream>>>(p, rp);
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
    RectParams rp;
    const int dpad = ctx->rect_dpad;
    const size_t nx = (size_t)p.tiles_x * p.V * dpad, ny = (size_t)p.tiles_y * p.V * dpad;
    rp.xt = (const uint32_t *)ctx->rect_tab.ptr;
    rp.yt = rp.xt + nx;
    rp.wt = rp.yt + ny;
    rp.lut = (const uint32_t *)ctx->fx_lut.ptr;
    rp.dpad = dpad;
    rp.slot_dw = ctx->rect_slot_dw;
    rp.ni = ctx->rect_ni;
    rp.instrs = ctx->rect_instrs;
    int slots = 3;
    if (const char *e = getenv("MVS_RECT_SLOTS")) slots = atoi(e);
    slots = max(2, min(slots, 8));
    rp.nslots = slots;
    size_t lds = (size_t)slots * rp.slot_dw * 4 + 1024;
    if (fused) lds = lds < 16384 ? 16384 : lds;  // the cross-wavefront depth selection borrows 16 KiB
    if (lds > 160 * 1024) return fail(ctx, MVS_EINVAL, "sweep_rect_launch: %zu bytes of LDS", lds);

    const int groups = div_up(p.tiles_x, 2) * div_up(p.tyn, 4);
    const int nch = p.chunk1 - p.chunk0, tiles = p.tiles_x * p.tyn;
    int want = (int)((flags >> 16) & 0xffu);
    if (!want) want = div_up(16 * ctx->num_cus, tiles);
    p.cps = div_up(nch, max(1, min(want, nch)));
    const int nsplit = div_up(nch, p.cps);
    int rc;
    if (fused && nsplit > 1) {
        if ((rc = ensure(ctx, ctx->best_parts, (size_t)nsplit * ctx->W * ctx->H * sizeof(uint2)))) return rc;
        p.part = (uint2 *)ctx->best_parts.ptr;
    }
    const dim3 grid((unsigned)(div_up(groups, 8) * 64), (unsigned)nsplit);
    rc = ctx->rect_rs == 96 ? launch_rect<96>(ctx, p, rp, grid, lds, vol, fused) : launch_rect<128>(ctx, p, rp, grid, lds, vol, fused);
    if (rc) return rc;
    return nsplit;
}

}  // namespace mvs
