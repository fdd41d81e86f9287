// v3_probe.hip -- what a "contract v3" sample loop for GENERAL cameras would cost on gfx950, measured before anything is built
// ("one bounded attempt at a real lever, or a final statement").
//
// Contract v2's sample (csrc/sweep_fx.hip) is 11.5 vector instructions: 7 FMAs + v_rcp_f32 for the projective position, v_perm + 2 shifts + 2
// masks for the two LDS addresses, v_dot4, v_sad.  The only way to fewer instructions per WAVE-sample is to amortise the position INSIDE a lane:
// anything that is uniform over a wavefront (a chord's end points per (row, plane, view), say) costs one instruction slot per wave-sample just
// like per-lane work does.  So the candidate is the TRANSPOSED mapping: a lane owns one (row, plane) of a 64 x 8 x 16 block and walks 32 columns;
// the position is exact at the two ends of its 32-column span (two evaluations of contract v2's arithmetic per 32 samples) and advances by one
// integer add per coordinate in 16.16 fixed point in between (a chord: off the hyperbola by < 1e-5 px over 32 px for the cameras of tracks/).
//
// Kernels (sample loops only: no region staging, no barriers, no epilogue -- the part of the kernel the contract decides):
//   A   contract v2 as shipped: lane = column, 2 rows x 16 planes per thread, per sample fma x3, rcp + Newton, fma x2, perm, shifts, masks,
//       2 LDS reads, dot4, sad (compiler-scheduled; the shipped loop is hand-pipelined and ~10 % faster than this form)
//   B0  transposed + chord: lane = (row, plane), 32 columns per thread, per sample 2 adds, perm, shifts, masks, 3 LDS reads (weights, quad,
//       255 x main pixel), dot4, sad; LDS rows 256 dwords apart as today (the two rows of a 32-lane group collide on a bank)
//   B1  the same with odd LDS rows skewed by 16 dwords (two more vector instructions per sample, no collision between the rows)
// Printed: ns per wave-sample and SIMD at 4 wavefronts per SIMD, the ratio to A, and what the ratio would make of the general kernel's
// sample-loop share.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/v3_probe tools/v3_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

constexpr int ROWS = 32, ROW_DW = 256, LUT_DW = 32, NV = 16, REPS = 8;
constexpr float MAGIC = 12582912.0f;

struct Views {
    float q[NV][12];
};

__device__ __forceinline__ float rcp_rn(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

__device__ __forceinline__ void fill_lds(uint32_t *lds)
{
    for (int i = threadIdx.x; i < ROWS * ROW_DW; i += 256) lds[i] = 0x01020304u * (uint32_t)(i * 2654435761u >> 24);
    __syncthreads();
}

// ---- A: contract v2's sample as shipped (lane = column; 2 rows x 16 planes per thread) ----
__global__ __launch_bounds__(256, 4) void loop_v2(Views V, const float *__restrict__ zc_g, uint32_t *__restrict__ out)
{
    __shared__ uint32_t lds[ROWS * ROW_DW];
    __shared__ uint32_t occupancy_pad[2048];   // 40 KB per workgroup: four workgroups per CU like the shipped kernel, not five
    fill_lds(lds);
    if (threadIdx.x == 0) occupancy_pad[blockIdx.x & 2047] = 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float xn = (2 * (blockIdx.x % 30 * 64 + lane) + 1) * (1.0f / 1920) - 1.0f;
    float zc[16];
    for (int k = 0; k < 16; k++) zc[k] = zc_g[k];
    uint32_t acc[2][16] = {};
    for (int rep = 0; rep < REPS; rep++)
        for (int v = 0; v < NV; v++) {
            const float *q = V.q[v];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float yn = 1.0f - (2 * (blockIdx.x / 30 * 8 + wave * 2 + j) + 1) * (1.0f / 1080);
                const float ax = __builtin_fmaf(q[0], xn, __builtin_fmaf(q[1], yn, q[3])), ay = __builtin_fmaf(q[4], xn, __builtin_fmaf(q[5], yn, q[7])),
                            aw = __builtin_fmaf(q[8], xn, __builtin_fmaf(q[9], yn, q[11]));
                const uint32_t im = 255u * (uint32_t)(lane + j);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const float z = zc[k];
                    const float sx = __builtin_fmaf(z, q[2], ax), sy = __builtin_fmaf(z, q[6], ay), sw = __builtin_fmaf(z, q[10], aw);
                    const float r = rcp_rn(sw);
                    const float Tx = __builtin_fmaf(sx, r, MAGIC + 4.0f), Ty = __builtin_fmaf(sy, r, MAGIC + 4.0f);
                    const uint32_t P = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, Ty), __builtin_bit_cast(uint32_t, Tx), 0x05010400u);
                    const uint32_t w = lds[((P >> 1) & 0x7c7cu) >> 2], quad = lds[(((P >> 14) & 0x7ffcu) >> 2) + LUT_DW];
                    acc[j][k] = __builtin_amdgcn_sad_u16(__builtin_amdgcn_udot4(quad, w, 0u, false), im, acc[j][k]);
                }
            }
        }
    uint32_t s = 0;
    for (int j = 0; j < 2; j++)
        for (int k = 0; k < 16; k++) s += acc[j][k];
    out[blockIdx.x * 256 + threadIdx.x] = s + occupancy_pad[threadIdx.x];
}

// ---- B: transposed mapping + chord positions (lane = (row, plane); 32 columns per thread) ----
template <bool SKEW>
__global__ __launch_bounds__(256, 4) void loop_v3(Views V, const float *__restrict__ zc_g, uint32_t *__restrict__ out)
{
    __shared__ uint32_t lds[ROWS * ROW_DW];
    __shared__ uint16_t im255[8][64];
    fill_lds(lds);
    for (int i = threadIdx.x; i < 8 * 64; i += 256) im255[i >> 6][i & 63] = (uint16_t)(255 * (i & 255));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int plane = lane & 15, rr = (lane >> 4) & 1, half = lane >> 5;
    const int row = wave * 2 + rr, col0 = half * 32;
    const float z = zc_g[plane];
    const float yn = 1.0f - (2 * (blockIdx.x / 30 * 8 + row) + 1) * (1.0f / 1080);
    const float xn0 = (2 * (blockIdx.x % 30 * 64 + col0) + 1) * (1.0f / 1920) - 1.0f, xn1 = xn0 + 64.0f / 1920;
    const uint16_t *imrow = im255[row] + col0;
    uint32_t acc[32] = {};
    for (int rep = 0; rep < REPS; rep++)
        for (int v = 0; v < NV; v++) {
            const float *q = V.q[v];
            // exact (contract v2) positions at the two ends of the thread's 32-column span, relative to the region, in 16.16 fixed point
            int U[2][2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float xn = e ? xn1 : xn0;
                const float sx = __builtin_fmaf(z, q[2], __builtin_fmaf(q[0], xn, __builtin_fmaf(q[1], yn, q[3])));
                const float sy = __builtin_fmaf(z, q[6], __builtin_fmaf(q[4], xn, __builtin_fmaf(q[5], yn, q[7])));
                const float sw = __builtin_fmaf(z, q[10], __builtin_fmaf(q[8], xn, __builtin_fmaf(q[9], yn, q[11])));
                const float r = rcp_rn(sw);
                const float Tx = __builtin_fmaf(sx, r, MAGIC + 4.0f), Ty = __builtin_fmaf(sy, r, MAGIC + 4.0f);
                U[e][0] = (int)((__builtin_bit_cast(uint32_t, Tx) & 0x3fffffu) << 8);   // 1/256 texel -> 16.16
                U[e][1] = (int)((__builtin_bit_cast(uint32_t, Ty) & 0x3fffffu) << 8);
            }
            int ux = U[0][0], uy = U[0][1];
            const int dux = (U[1][0] - U[0][0]) >> 5, duy = (U[1][1] - U[0][1]) >> 5;
#pragma unroll
            for (int c = 0; c < 32; c++) {
                const uint32_t P = __builtin_amdgcn_perm((uint32_t)uy, (uint32_t)ux, 0x06020501u);  // [iy : ix : fy8 : fx8]
                uint32_t ta = ((P >> 14) & 0x7ffcu);
                if (SKEW) ta += (P >> 18) & 0x40u;  // odd LDS rows start 16 dwords later (bit 24 of P = iy & 1 -> byte offset 64)
                const uint32_t w = lds[((P >> 1) & 0x7c7cu) >> 2], quad = lds[(ta >> 2) + LUT_DW];
                acc[c] = __builtin_amdgcn_sad_u16(__builtin_amdgcn_udot4(quad, w, 0u, false), (uint32_t)imrow[c], acc[c]);
                ux += dux;
                uy += duy;
            }
        }
    uint32_t s = 0;
    for (int c = 0; c < 32; c++) s += acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    // a ring of general cameras as bench.py --general-cameras builds them, reduced to what the loops need: view matrices whose positions stay
    // inside a 224 x 32 quad region for a 64 x 8 tile over 16 planes (parallax ~1.3 quads per plane, 12 mrad of rotation)
    Views V;
    for (int v = 0; v < NV; v++) {
        const float a = 6.2831853f * v / NV, par = 52.0f * __builtin_cosf(a), pary = 16.0f * __builtin_sinf(a);   // 1.3 / 0.4 quads per plane
        const float rot = 0.012f;
        float *q = V.q[v];
        // one quad per column and per row (xn spans 2 over 1920 columns, yn over 1080 rows), a small rotation, w = (1 + 0.15 z + ...) / 256
        q[0] = 960.0f;       q[1] = rot * 540;    q[2] = par;  q[3] = 960.0f + 60;
        q[4] = -rot * 960;   q[5] = -540.0f;      q[6] = pary; q[7] = 540.0f + 12;
        q[8] = rot * 0.05f / 256;  q[9] = -rot * 0.05f / 256; q[10] = 0.15f / 256; q[11] = 1.0f / 256;
    }
    std::vector<float> zc(16);
    for (int k = 0; k < 16; k++) zc[k] = -0.2f + 0.025f * k;
    float *zd;
    uint32_t *out;
    const int blocks = 256 * 4 * 4;
    CHECK(hipMalloc(&zd, 64));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(zd, zc.data(), 64, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double wave_samples = (double)blocks * 4 * REPS * NV * 32;  // per launch: every wavefront does 32 samples per lane and view
    auto time = [&](const char *name, auto launch, double base) {
        for (int i = 0; i < 3; i++) launch();
        CHECK(hipEventRecord(e0));
        const int n = 10;
        for (int i = 0; i < n; i++) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double ns = ms / n * 1e6 / (wave_samples / 1024.0);  // per wave-sample and SIMD
        printf("%-44s %7.3f ms per launch  %6.2f ns per wave-sample and SIMD  (%.2f x A)\n", name, ms / n, ns, base > 0 ? ns / base : 1.0);
        return ns;
    };
    const double a = time("A  contract v2, lane = column", [&] { loop_v2<<<blocks, 256>>>(V, zd, out); }, 0);
    const double b0 = time("B0 transposed + chord, rows 256 dwords apart", [&] { loop_v3<false><<<blocks, 256>>>(V, zd, out); }, a);
    const double b1 = time("B1 transposed + chord, odd rows skewed", [&] { loop_v3<true><<<blocks, 256>>>(V, zd, out); }, a);
    CHECK(hipDeviceSynchronize());
    // c3: 66.4 M wave-samples on 1024 SIMDs
    printf("c3's 66.4 M wave-samples at these rates: A %.3f ms, B0 %.3f ms, B1 %.3f ms (sample loops alone; sweep_fx_tiled runs 1.67 ms, of which ~1.0 ms is there without any sampling)\n",
           66.4e6 / 1024 * a * 1e-6, 66.4e6 / 1024 * b0 * 1e-6, 66.4e6 / 1024 * b1 * 1e-6);
    return 0;
}
