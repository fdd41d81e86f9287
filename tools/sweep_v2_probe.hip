// sweep_v2_probe.hip -- design probe for the fixed-point sweep sampler (round 2).
// Build: hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 tools/sweep_v2_probe.hip -o tools/sweep_v2_probe
// 1. issue rate of the integer / select / compare instructions the fixed-point sampler is built from
// 2. v_mfma_f32_32x32x2_f32: order of accumulation over k, rounding (== one fmaf per k?), denormal results, issue rate
//    alone and beside VALU work
// 3. ns per wave-sample per SIMD of candidate inner loops (synthetic coordinates, 64 KB of LDS per workgroup)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

constexpr int ITERS = 2000;
constexpr int UNROLL = 16;

// ---------------------------------------------------------------------------------------------- 1. issue rates
template <int OP>
__global__ void probe(float *out, unsigned long long *cyc, float seed)
{
    float a[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) a[i] = seed + i + threadIdx.x;
    const float c0 = seed * 1.0001f, c1 = seed * 0.37f;
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(c0), "v"((float)(threadIdx.x & 1)) : "vcc");
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 2) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 4) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 5) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 6) asm volatile("v_bfe_u32 %0, %0, 6, 8" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c0) : );
            if (OP == 8) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 9) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");
            if (OP == 10) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 11) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 12) asm volatile("v_alignbit_b32 %0, %0, %1, 24" : "+v"(a[i]) : "v"(c0));
            if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c0));
            if (OP == 14) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 15) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 16) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
            if (OP == 17) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[i]));
            if (OP == 18) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 19) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 20) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");
            if (OP == 21) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 22) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(c0));
            if (OP == 23) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 24) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, float *out, unsigned long long *cyc)
{
    printf("%-20s", name);
    double prev_ms = 0;
    int prev_w = 0;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        probe<OP><<<blocks, 256>>>(out, cyc, 1.25f);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        probe<OP><<<blocks, 256>>>(out, cyc, 1.25f);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  w%d %.3f ms", wps, ms);
        if (wps == 8) printf("   => %.2f ns per wave-instr per SIMD (slope w4->w8)", (ms - prev_ms) * 1e6 / ((double)ITERS * UNROLL * (wps - prev_w)));
        prev_ms = ms; prev_w = wps;
    }
    printf("\n");
}

// ---------------------------------------------------------------------------------------------- 2. MFMA semantics
// D[i][j] = A[i][0]*B[0][j] + A[i][1]*B[1][j]; lane l supplies A[l%32][l/32] and B[l/32][l%32];
// lane l receives rows i = 8*(r/4) + 4*(l/32) + r%4 of column j = l%32.
__global__ void mfma_semantics(const float *a0, const float *a1, const float *b0, const float *b1, float *d)
{
    const int l = threadIdx.x;
    const float av = l < 32 ? a0[l] : a1[l - 32];
    const float bv = l < 32 ? b0[l] : b1[l - 32];
    f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = 8 * (r / 4) + 4 * (l / 32) + (r % 4), j = l % 32;
        d[i * 32 + j] = c[r];
    }
}

static void test_mfma_semantics()
{
    float *da0, *da1, *db0, *db1, *dd;
    CHECK(hipMalloc(&da0, 128)); CHECK(hipMalloc(&da1, 128)); CHECK(hipMalloc(&db0, 128)); CHECK(hipMalloc(&db1, 128));
    CHECK(hipMalloc(&dd, 4096));
    std::vector<float> a0(32), a1(32), b0(32), b1(32), d(1024);
    srand(7);
    auto rnd = [] { return (float)rand() / (float)RAND_MAX; };
    for (int mode = 0; mode < 5; mode++) {
        long n_k01 = 0, n_k10 = 0, n_prod_rounded = 0, n_other = 0, n_total = 0;
        for (int rep = 0; rep < 200; rep++) {
            for (int i = 0; i < 32; i++) {
                if (mode == 0) {  // generic values
                    a0[i] = rnd() * 2 - 1; a1[i] = rnd() * 2 - 1; b0[i] = rnd() * 100; b1[i] = rnd() * 100 - 50;
                } else if (mode == 1) {  // the intended use: 1 * intercept + z * slope
                    a0[i] = 1.0f; a1[i] = rnd() * 2 - 1; b0[i] = rnd() * 2000; b1[i] = rnd() * 200 - 100;
                } else if (mode == 2) {  // swapped roles: z * slope first, 1 * intercept second
                    a1[i] = 1.0f; a0[i] = rnd() * 2 - 1; b1[i] = rnd() * 2000; b0[i] = rnd() * 200 - 100;
                } else if (mode == 3) {  // magic-number results: intercept carries 1.5 * 2^23
                    a0[i] = 1.0f; a1[i] = rnd() * 2 - 1; b0[i] = 12582912.0f + floorf(rnd() * 60000); b1[i] = rnd() * 20000 - 10000;
                } else {  // denormal results
                    a0[i] = 1.0f; a1[i] = rnd() * 2 - 1; b0[i] = ldexpf(floorf(rnd() * 60000), -149); b1[i] = ldexpf(rnd() * 20000 - 10000, -149);
                }
            }
            CHECK(hipMemcpy(da0, a0.data(), 128, hipMemcpyHostToDevice)); CHECK(hipMemcpy(da1, a1.data(), 128, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(db0, b0.data(), 128, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db1, b1.data(), 128, hipMemcpyHostToDevice));
            mfma_semantics<<<1, 64>>>(da0, da1, db0, db1, dd);
            CHECK(hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost));
            for (int i = 0; i < 32; i++)
                for (int j = 0; j < 32; j++) {
                    const float got = d[i * 32 + j];
                    const float k01 = fmaf(a1[i], b1[j], fmaf(a0[i], b0[j], 0.0f));  // k = 0 first, one rounding per k
                    const float k10 = fmaf(a0[i], b0[j], fmaf(a1[i], b1[j], 0.0f));  // k = 1 first
                    volatile float p0 = a0[i] * b0[j], p1 = a1[i] * b1[j];
                    const float pr = p0 + p1;  // both products rounded, then added
                    n_total++;
                    uint32_t g, e1, e2, e3;
                    memcpy(&g, &got, 4); memcpy(&e1, &k01, 4); memcpy(&e2, &k10, 4); memcpy(&e3, &pr, 4);
                    if (g == e1) n_k01++;
                    if (g == e2) n_k10++;
                    if (g == e3) n_prod_rounded++;
                    if (g != e1 && g != e2 && g != e3) {
                        n_other++;
                        if (n_other <= 3) printf("   other: a0 %a b0 %a a1 %a b1 %a got %a k01 %a k10 %a\n", a0[i], b0[j], a1[i], b1[j], got, k01, k10);
                    }
                }
        }
        printf("mfma 32x32x2 f32 mode %d: of %ld results  ==fma(k1,fma(k0)) %ld  ==fma(k0,fma(k1)) %ld  ==rounded products %ld  none %ld\n", mode, n_total,
               n_k01, n_k10, n_prod_rounded, n_other);
    }
}

// ---------------------------------------------------------------------------------------------- 3. candidate loops
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t acc)
{
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
__device__ __forceinline__ uint32_t sad_u16(uint32_t a, uint32_t b, uint32_t acc)
{
    uint32_t d;
    asm("v_sad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
__device__ __forceinline__ float rcp_rn(float w)
{
    const float r0 = __builtin_amdgcn_rcpf(w);
    const float e = __builtin_fmaf(-w, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}

constexpr int NPX = 4, NPL = 16;        // pixel rows and planes per lane
constexpr float MAGIC = 12582912.0f;    // 1.5 * 2^23: ulp 1

// the fixed-point tail: (Tx, Ty) as magic floats -> perm -> two LDS reads -> dot4 -> sad_u16
// TAILMODE bit 0: no LUT read (weights = address bits), bit 1: no tile read, bit 2: LUT from global memory (L1) instead of LDS
__device__ const char *g_lut;
template <int TAILMODE>
__device__ __forceinline__ uint32_t tail_fx_t(float Tx, float Ty, const char *lds, uint32_t Im255, uint32_t acc, const char *glut)
{
    const uint32_t P = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, Ty), __builtin_bit_cast(uint32_t, Tx), 0x05010400u);
    const uint32_t la = P & 0xfcfcu;
    const uint32_t ta = (P >> 14) & 0xfffcu;
    uint32_t w, q;
    if (TAILMODE & 1) w = la; else if (TAILMODE & 4) w = *(const uint32_t *)(glut + la); else w = *(const uint32_t *)(lds + la);
    if (TAILMODE & 2) q = ta; else q = *(const uint32_t *)(lds + ta + 256);
    const uint32_t dot = __builtin_amdgcn_udot4(q, w, 0u, false);
    return sad_u16(dot, Im255, acc);
}
__device__ __forceinline__ uint32_t tail_fx(float Tx, float Ty, const char *lds, uint32_t Im255, uint32_t acc)
{
    return tail_fx_t<0>(Tx, Ty, lds, Im255, acc, nullptr);
}

// VAR1 with the tail's memory sources switched off / moved, and scalar instead of packed FMAs (SCALAR)
template <int TAILMODE, bool SCALAR, int LDSB>
__global__ __launch_bounds__(256, 2) void tail_probe(float *out, int nviews, float seed, const char *glut)
{
    __shared__ __attribute__((aligned(16))) char lds[LDSB];
    for (int i = threadIdx.x; i < LDSB / 4; i += 256) ((uint32_t *)lds)[i] = (uint32_t)i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, ph = lane >> 5;
    uint32_t acc[4][16];
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[p][k] = 0u;
    float z[16];
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = -1.0f + (float)(8 * (r / 4) + 4 * ph + (r % 4)) * (1.0f / 64.0f) + seed * 1e-3f;
    uint32_t Im[4];
#pragma unroll
    for (int p = 0; p < 4; p++) Im[p] = (uint32_t)((j * 7 + p * 13) & 255) * 255u;
    for (int v = 0; v < nviews; v++) {
        const float bx = 9.0f + 0.01f * v, by = 2.0f + 0.02f * v;
        const float q0 = 1.0f + 0.001f * v, offx = 12582912.0f + 2.0f, offy = 12582912.0f + 2.0f;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const float Ax = __builtin_fmaf(q0, (float)j, 20.0f + seed + 0.3f * (float)p), Ay = __builtin_fmaf(0.01f * q0, (float)j, 10.0f + (float)p), Aw = 1.0f + 1e-4f * (float)j;
            const float r256 = rcp_rn(Aw) * 256.0f;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                if (SCALAR) {
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const float sx = __builtin_fmaf(z[k + i], bx, Ax), sy = __builtin_fmaf(z[k + i], by, Ay);
                        const float Tx = __builtin_fmaf(sx, r256, offx), Ty = __builtin_fmaf(sy, r256, offy);
                        acc[p][k + i] = tail_fx_t<TAILMODE>(Tx, Ty, lds, Im[p], acc[p][k + i], glut);
                    }
                } else {
                    const f32x2 zz = {z[k], z[k + 1]};
                    const f32x2 sx = __builtin_elementwise_fma(zz, (f32x2)(bx), (f32x2)(Ax));
                    const f32x2 sy = __builtin_elementwise_fma(zz, (f32x2)(by), (f32x2)(Ay));
                    const f32x2 Tx = __builtin_elementwise_fma(sx, (f32x2)(r256), (f32x2)(offx));
                    const f32x2 Ty = __builtin_elementwise_fma(sy, (f32x2)(r256), (f32x2)(offy));
                    acc[p][k] = tail_fx_t<TAILMODE>(Tx.x, Ty.x, lds, Im[p], acc[p][k], glut);
                    acc[p][k + 1] = tail_fx_t<TAILMODE>(Tx.y, Ty.y, lds, Im[p], acc[p][k + 1], glut);
                }
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int k = 0; k < 16; k++) s += acc[p][k];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}


// Prototype of the real inner loop: lane = pixel column, NPXP pixel rows x 32 planes per lane, plane constants in SGPRs,
// samples processed in groups of GS with the LDS reads of group g+1 in flight during the dot4/sad of group g (inline asm
// reads, one counted wait per group).  GENERAL: per-sample reciprocal (v_rcp_f32 + Newton) instead of a hoisted one.
template <int GS, bool GENERAL, int NPXP, int LDSB, int WAVES_PER_EU>
__global__ __launch_bounds__(256, WAVES_PER_EU) void pipe_probe(float *out, int nviews, float seed, const float *ztab)
{
    __shared__ __attribute__((aligned(16))) char lds[LDSB];
    for (int i = threadIdx.x; i < LDSB / 4; i += 256) ((uint32_t *)lds)[i] = (uint32_t)i * 2654435761u;
    __syncthreads();
    constexpr int PCN = 32;
    const int lane = threadIdx.x & 63;
    uint32_t acc[NPXP][PCN];
#pragma unroll
    for (int p = 0; p < NPXP; p++)
#pragma unroll
        for (int k = 0; k < PCN; k++) acc[p][k] = 0u;
    float zc[PCN];
#pragma unroll
    for (int k = 0; k < PCN; k++) zc[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ztab[k])));
    uint32_t Im[NPXP];
#pragma unroll
    for (int p = 0; p < NPXP; p++) Im[p] = (uint32_t)((lane * 7 + p * 13) & 255) * 255u;
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    for (int v = 0; v < nviews; v++) {
        const float bx = 9.0f + 0.01f * v, by = 2.0f + 0.02f * v, bw = 0.001f * v;
        const float q0 = 1.0f + 0.001f * v, offx = 12582912.0f + 2.0f, offy = 12582912.0f + 2.0f;
#pragma unroll
        for (int p = 0; p < NPXP; p++) {
            const float Ax = __builtin_fmaf(q0, (float)lane, 20.0f + seed + 0.3f * (float)p), Ay = __builtin_fmaf(0.01f * q0, (float)lane, 10.0f + (float)p), Aw = 1.0f + 1e-4f * (float)lane;
            const float r256c = rcp_rn(Aw) * 256.0f;
            uint32_t la[2][GS], ta[2][GS];
            auto address_stage = [&](int g, int buf) {
#pragma unroll
                for (int i = 0; i < GS; i++) {
                    const float z = zc[g * GS + i];
                    const float sx = __builtin_fmaf(z, bx, Ax), sy = __builtin_fmaf(z, by, Ay);
                    float r256 = r256c;
                    if (GENERAL) {
                        const float sw = __builtin_fmaf(z, bw, Aw);
                        const float r0 = __builtin_amdgcn_rcpf(sw);
                        const float e = __builtin_fmaf(-sw, r0, 1.0f);
                        r256 = __builtin_fmaf(e, r0, r0) * 256.0f;
                    }
                    const float Tx = __builtin_fmaf(sx, r256, offx), Ty = __builtin_fmaf(sy, r256, offy);
                    const uint32_t P = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, Ty), __builtin_bit_cast(uint32_t, Tx), 0x05010400u);
                    la[buf][i] = (P & 0xfcfcu) + lds_base;
                    ta[buf][i] = ((P >> 14) & 0xfffcu) + lds_base;
                }
            };
            auto issue_reads = [&](int buf) {
#pragma unroll
                for (int i = 0; i < GS; i++) {
                    asm volatile("ds_read_b32 %0, %0" : "+v"(la[buf][i]));
                    asm volatile("ds_read_b32 %0, %0 offset:256" : "+v"(ta[buf][i]));
                }
            };
            address_stage(0, 0);
            issue_reads(0);
#pragma unroll
            for (int g = 0; g < PCN / GS; g++) {
                const int buf = g & 1;
                if (g + 1 < PCN / GS) address_stage(g + 1, buf ^ 1);
                // wait for group g's reads; the next group's addresses are operands too so that stage cannot sink below
                if (GS == 8)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(la[buf][0]), "+v"(la[buf][1]), "+v"(la[buf][2]), "+v"(la[buf][3]), "+v"(la[buf][4]), "+v"(la[buf][5]), "+v"(la[buf][6]), "+v"(la[buf][7]),
                                 "+v"(ta[buf][0]), "+v"(ta[buf][1]), "+v"(ta[buf][2]), "+v"(ta[buf][3]), "+v"(ta[buf][4]), "+v"(ta[buf][5]), "+v"(ta[buf][6]), "+v"(ta[buf][7]),
                                 "+v"(la[buf ^ 1][0]), "+v"(ta[buf ^ 1][0]), "+v"(la[buf ^ 1][GS - 1]), "+v"(ta[buf ^ 1][GS - 1]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(la[buf][0]), "+v"(la[buf][1]), "+v"(la[buf][2]), "+v"(la[buf][3]),
                                 "+v"(ta[buf][0]), "+v"(ta[buf][1]), "+v"(ta[buf][2]), "+v"(ta[buf][3]),
                                 "+v"(la[buf ^ 1][0]), "+v"(ta[buf ^ 1][0]), "+v"(la[buf ^ 1][GS - 1]), "+v"(ta[buf ^ 1][GS - 1]));
                if (g + 1 < PCN / GS) issue_reads(buf ^ 1);
#pragma unroll
                for (int i = 0; i < GS; i++) {
                    const uint32_t dot = __builtin_amdgcn_udot4(ta[buf][i], la[buf][i], 0u, false);
                    acc[p][g * GS + i] = sad_u16(dot, Im[p], acc[p][g * GS + i]);
                }
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int p = 0; p < NPXP; p++)
#pragma unroll
        for (int k = 0; k < PCN; k++) s += acc[p][k];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

template <int GS, bool GENERAL, int NPXP, int LDSB, int WAVES_PER_EU>
static void run_pipe(const char *name, float *out, const float *ztab)
{
    const int nviews = 256;
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pipe_probe<GS, GENERAL, NPXP, LDSB, WAVES_PER_EU>, 256, 0));
    const int blocks = 256 * per_cu * 4;
    pipe_probe<GS, GENERAL, NPXP, LDSB, WAVES_PER_EU><<<blocks, 256>>>(out, nviews, 1.0f, ztab);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    pipe_probe<GS, GENERAL, NPXP, LDSB, WAVES_PER_EU><<<blocks, 256>>>(out, nviews, 1.0f, ztab);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_samples = (double)blocks * 4 * nviews * NPXP * 32;
    printf("%-52s lds %3d KB  %d wg/CU  %.3f ms  -> %.2f ns per wave-sample per SIMD\n", name, LDSB / 1024, per_cu, ms, ms * 1e6 * 1024.0 / wave_samples);
}

template <int TAILMODE, bool SCALAR, int LDSB>
static void run_tail(const char *name, float *out, const char *glut)
{
    const int nviews = 256;
    int per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tail_probe<TAILMODE, SCALAR, LDSB>, 256, 0));
    const int blocks = 256 * per_cu * 4;
    tail_probe<TAILMODE, SCALAR, LDSB><<<blocks, 256>>>(out, nviews, 1.0f, glut);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    tail_probe<TAILMODE, SCALAR, LDSB><<<blocks, 256>>>(out, nviews, 1.0f, glut);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_samples = (double)blocks * 4 * nviews * 64;
    printf("%-52s lds %3d KB  %d wg/CU  %.3f ms  -> %.2f ns per wave-sample per SIMD\n", name, LDSB / 1024, per_cu, ms, ms * 1e6 * 1024.0 / wave_samples);
}

// LDS gather rate: ds_read_b32 with per-lane addresses of a given pattern, 16 reads in flight
template <int PATTERN>
__global__ void lds_gather(float *out, int seed)
{
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((uint32_t *)lds)[i] = (uint32_t)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t addr;
    if (PATTERN == 0) addr = 0;                                       // all lanes one address (broadcast)
    if (PATTERN == 1) addr = (lane & 31) * 4 + (lane >> 5) * 1024;    // consecutive dwords per half wave
    if (PATTERN == 2) addr = ((lane * 2654435761u) >> 17) & 0x7ffc;   // random
    if (PATTERN == 3) addr = (lane >> 3) * 4;                         // 8 distinct addresses
    if (PATTERN == 4) addr = (lane & 31) * 4 + ((lane & 31) > 20 ? 1024 : 0) + (lane >> 5) * 2048;  // a row step inside the half wave
    if (PATTERN == 5) addr = lane * 2;                                // 2-byte stride, overlapping dwords (a u16 column-pair layout)
    if (PATTERN == 6) addr = lane * 2 + 2;                            // the same, odd start
    if (PATTERN == 7) addr = lane * 4 + 2;                            // dword stride, every read misaligned by 2
    addr += seed;
    uint32_t acc = 0;
    for (int it = 0; it < ITERS; it++) {
        uint32_t v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "n"(i * 128));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("" ::"v"(v[i]));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}

template <int PATTERN>
static void run_gather(const char *name, float *out)
{
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        lds_gather<PATTERN><<<blocks, 256>>>(out, 0);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        lds_gather<PATTERN><<<blocks, 256>>>(out, 0);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("ds_read_b32 %-28s w%d: %.3f ms -> %.2f ns per wave-instr per CU\n", name, wps, ms, ms * 1e6 / ((double)ITERS * 16 * wps * 4));
    }
}

// VAR 0: round-1 arithmetic (plane-independent w): float coordinates, fract/cvt, f16 quads, 3 FMAs, cvt, sad_u32
// VAR 1: fixed-point sampler, coordinates by VALU (s = fma(z, B, A) packed over plane pairs, T = fma(s, r256, off))
// VAR 2: fixed-point sampler, s by MFMA (sx, sy), T = fma(s, r256, off) by VALU        (plane-independent w)
// VAR 3: fixed-point sampler, T directly by MFMA (no per-sample coordinate VALU)            (plane-independent w, contract branch)
// VAR 4: fixed-point sampler, general cameras: s by VALU, rcp + Newton, T by VALU
// VAR 5: fixed-point sampler, general cameras: sx, sy, sw by MFMA, rcp + Newton, T by VALU
// VAR 6: as 5 with the reciprocal started from an MFMA-interpolated guess (no v_rcp_f32)
template <int VAR, int LDSB>
__global__ __launch_bounds__(256, 2) void loop_probe(float *out, int nviews, float seed)
{
    __shared__ __attribute__((aligned(16))) char lds[LDSB];
    for (int i = threadIdx.x; i < LDSB / 4; i += 256) ((uint32_t *)lds)[i] = (uint32_t)i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, ph = lane >> 5;
    uint32_t acc[NPX][NPL];
#pragma unroll
    for (int p = 0; p < NPX; p++)
#pragma unroll
        for (int k = 0; k < NPL; k++) acc[p][k] = 0u;
    float z[NPL];
#pragma unroll
    for (int r = 0; r < NPL; r++) z[r] = -1.0f + (float)(8 * (r / 4) + 4 * ph + (r % 4)) * (1.0f / 64.0f) + seed * 1e-3f;
    const float zlane = lane < 32 ? 1.0f : -1.0f + (float)(lane - 32) * (1.0f / 64.0f);  // MFMA A operand: k = 0 -> 1, k = 1 -> z_i
    uint32_t Im[NPX];
#pragma unroll
    for (int p = 0; p < NPX; p++) Im[p] = (uint32_t)((j * 7 + p * 13) & 255) * (VAR == 0 ? 1u : 255u);
    for (int v = 0; v < nviews; v++) {
        // "view matrix": wave-uniform values that change per view
        const float bx = 9.0f + 0.01f * v, by = 2.0f + 0.02f * v, bw = 0.001f * v;
        const float q0 = 1.0f + 0.001f * v, offx = MAGIC + 2.0f - 256.0f * 0.0f, offy = MAGIC + 2.0f;
#pragma unroll
        for (int p = 0; p < NPX; p++) {
            // per (pixel, view) set-up
            const float Ax = __builtin_fmaf(q0, (float)j, 20.0f + seed + 0.3f * (float)p), Ay = __builtin_fmaf(0.01f * q0, (float)j, 10.0f + (float)p), Aw = 1.0f + 1e-4f * (float)j;
            if (VAR == 0) {
                const float r = rcp_rn(Aw);
                const int negorg = 64 - (int)seed;
#pragma unroll
                for (int k = 0; k < NPL; k += 2) {
                    const f32x2 zz = {z[k], z[k + 1]};
                    const f32x2 sx = __builtin_elementwise_fma(zz, (f32x2)(bx), (f32x2)(Ax));
                    const f32x2 sy = __builtin_elementwise_fma(zz, (f32x2)(by), (f32x2)(Ay));
                    const f32x2 cx = sx * (f32x2)(r), cy = sy * (f32x2)(r);
                    f32x2 fy, a, b;
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const float fx = __builtin_amdgcn_fractf(cx[i]);
                        fy[i] = __builtin_amdgcn_fractf(cy[i]);
                        const int ix = (int)cx[i], iy = (int)cy[i];
                        int row_off, off;
                        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(row_off) : "v"(iy), "s"(1024), "v"(negorg));
                        asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(off) : "v"(ix), "v"(row_off));
                        const half4_t h = *(const half4_t *)(lds + (off & 0xfff8));
                        a[i] = __builtin_fmaf(fx, (float)h[1], (float)h[0]);
                        b[i] = __builtin_fmaf(fx, (float)h[3], (float)h[2]);
                    }
                    const f32x2 res = __builtin_elementwise_fma(fy, b, a);
                    acc[p][k] = sad_u32((uint32_t)(int)res.x, Im[p], acc[p][k]);
                    acc[p][k + 1] = sad_u32((uint32_t)(int)res.y, Im[p], acc[p][k + 1]);
                }
            } else if (VAR == 1 || VAR == 4) {
                float r256c = 0.f;
                if (VAR == 1) r256c = rcp_rn(Aw) * 256.0f;
#pragma unroll
                for (int k = 0; k < NPL; k += 2) {
                    const f32x2 zz = {z[k], z[k + 1]};
                    const f32x2 sx = __builtin_elementwise_fma(zz, (f32x2)(bx), (f32x2)(Ax));
                    const f32x2 sy = __builtin_elementwise_fma(zz, (f32x2)(by), (f32x2)(Ay));
                    f32x2 r256;
                    if (VAR == 1) {
                        r256 = (f32x2)(r256c);
                    } else {
                        const f32x2 sw = __builtin_elementwise_fma(zz, (f32x2)(bw), (f32x2)(Aw));
                        const f32x2 r0 = {__builtin_amdgcn_rcpf(sw.x), __builtin_amdgcn_rcpf(sw.y)};
                        const f32x2 e = __builtin_elementwise_fma(-sw, r0, (f32x2)(1.0f));
                        r256 = __builtin_elementwise_fma(e, r0, r0) * (f32x2)(256.0f);
                    }
                    const f32x2 Tx = __builtin_elementwise_fma(sx, r256, (f32x2)(offx));
                    const f32x2 Ty = __builtin_elementwise_fma(sy, r256, (f32x2)(offy));
                    acc[p][k] = tail_fx(Tx.x, Ty.x, lds, Im[p], acc[p][k]);
                    acc[p][k + 1] = tail_fx(Tx.y, Ty.y, lds, Im[p], acc[p][k + 1]);
                }
            } else {
                f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (VAR == 3) {
                    const float r256 = rcp_rn(Aw) * 256.0f;
                    const float opx = lane < 32 ? __builtin_fmaf(Ax, r256, offx) : bx * r256;
                    const float opy = lane < 32 ? __builtin_fmaf(Ay, r256, offy) : by * r256;
                    const f32x16 Tx = __builtin_amdgcn_mfma_f32_32x32x2f32(zlane, opx, zero, 0, 0, 0);
                    const f32x16 Ty = __builtin_amdgcn_mfma_f32_32x32x2f32(zlane, opy, zero, 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < NPL; k++) acc[p][k] = tail_fx(Tx[k], Ty[k], lds, Im[p], acc[p][k]);
                } else {
                    const float opx = lane < 32 ? Ax : bx, opy = lane < 32 ? Ay : by, opw = lane < 32 ? Aw : bw;
                    const f32x16 sx = __builtin_amdgcn_mfma_f32_32x32x2f32(zlane, opx, zero, 0, 0, 0);
                    const f32x16 sy = __builtin_amdgcn_mfma_f32_32x32x2f32(zlane, opy, zero, 0, 0, 0);
                    if (VAR == 2) {
                        const float r256 = rcp_rn(Aw) * 256.0f;
#pragma unroll
                        for (int k = 0; k < NPL; k += 2) {
                            const f32x2 sxp = {sx[k], sx[k + 1]}, syp = {sy[k], sy[k + 1]};
                            const f32x2 Tx = __builtin_elementwise_fma(sxp, (f32x2)(r256), (f32x2)(offx));
                            const f32x2 Ty = __builtin_elementwise_fma(syp, (f32x2)(r256), (f32x2)(offy));
                            acc[p][k] = tail_fx(Tx.x, Ty.x, lds, Im[p], acc[p][k]);
                            acc[p][k + 1] = tail_fx(Tx.y, Ty.y, lds, Im[p], acc[p][k + 1]);
                        }
                    } else {
                        const f32x16 sw = __builtin_amdgcn_mfma_f32_32x32x2f32(zlane, opw, zero, 0, 0, 0);
                        f32x16 rg = zero;
                        if (VAR == 6) {
                            const float opr = lane < 32 ? rcp_rn(Aw) : -bw;
                            rg = __builtin_amdgcn_mfma_f32_32x32x2f32(zlane, opr, zero, 0, 0, 0);
                        }
#pragma unroll
                        for (int k = 0; k < NPL; k += 2) {
                            const f32x2 sxp = {sx[k], sx[k + 1]}, syp = {sy[k], sy[k + 1]}, swp = {sw[k], sw[k + 1]};
                            f32x2 r0;
                            if (VAR == 6) {
                                r0 = (f32x2){rg[k], rg[k + 1]};
                            } else {
                                r0 = (f32x2){__builtin_amdgcn_rcpf(swp.x), __builtin_amdgcn_rcpf(swp.y)};
                            }
                            const f32x2 e = __builtin_elementwise_fma(-swp, r0, (f32x2)(1.0f));
                            const f32x2 r256 = __builtin_elementwise_fma(e, r0, r0) * (f32x2)(256.0f);
                            const f32x2 Tx = __builtin_elementwise_fma(sxp, r256, (f32x2)(offx));
                            const f32x2 Ty = __builtin_elementwise_fma(syp, r256, (f32x2)(offy));
                            acc[p][k] = tail_fx(Tx.x, Ty.x, lds, Im[p], acc[p][k]);
                            acc[p][k + 1] = tail_fx(Tx.y, Ty.y, lds, Im[p], acc[p][k + 1]);
                        }
                    }
                }
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int p = 0; p < NPX; p++)
#pragma unroll
        for (int k = 0; k < NPL; k++) s += acc[p][k];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

template <int VAR, int LDSB>
static void run_loop(const char *name, float *out)
{
    const int lds_bytes = LDSB;
    const int nviews = 256;
    for (int rounds : {4}) {
        int per_cu = 0;
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, loop_probe<VAR, LDSB>, 256, 0));
        const int blocks = 256 * per_cu * rounds;
        loop_probe<VAR, LDSB><<<blocks, 256>>>(out, nviews, 1.0f);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        loop_probe<VAR, LDSB><<<blocks, 256>>>(out, nviews, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_samples = (double)blocks * 4 * nviews * NPX * NPL;
        printf("%-44s lds %3d KB  %d wg/CU  %.3f ms  -> %.2f ns per wave-sample per SIMD\n", name, lds_bytes / 1024, per_cu, ms, ms * 1e6 * 1024.0 / wave_samples);
    }
}

// MFMA issue rate alone and beside VALU work
template <int NVALU>
__global__ void mfma_rate(float *out, float seed)
{
    f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + i;
    const float av = seed + threadIdx.x, bv = seed * 0.5f;
    for (int it = 0; it < ITERS; it++) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NVALU; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i % 8]) : "v"(av), "v"(bv));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += c[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NVALU>
static void run_mfma_rate(float *out)
{
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        mfma_rate<NVALU><<<blocks, 256>>>(out, 1.0f);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        mfma_rate<NVALU><<<blocks, 256>>>(out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("mfma_f32_32x32x2 + %2d v_fma per iteration, %d waves/SIMD: %.3f ms -> %.1f ns per iteration per SIMD\n", NVALU, wps, ms, ms * 1e6 / ((double)ITERS * wps));
    }
}

int main(int argc, char **argv)
{
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 256 * 64 * 256 * 4));
    CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    if (argc > 1 && argv[1][0] == 'g') {  // LDS gather patterns only
        run_gather<1>("consecutive per half wave", out);
        run_gather<5>("2-byte stride, even start", out);
        run_gather<6>("2-byte stride, odd start", out);
        run_gather<7>("dword stride, misaligned by 2", out);
        return 0;
    }
    {
        char *glut;
        CHECK(hipMalloc(&glut, 65536));
        CHECK(hipMemset(glut, 0x11, 65536));
        run_tail<0, false, 65536>("tail: LUT lds, tile lds, packed fma", out, glut);
        run_tail<0, true, 65536>("tail: LUT lds, tile lds, scalar fma", out, glut);
        run_tail<1, true, 65536>("tail: no LUT read, tile lds, scalar fma", out, glut);
        run_tail<2, true, 65536>("tail: LUT lds, no tile read, scalar fma", out, glut);
        run_tail<3, true, 65536>("tail: no reads at all (pure VALU), scalar fma", out, glut);
        run_tail<3, false, 65536>("tail: no reads at all (pure VALU), packed fma", out, glut);
        run_tail<4, true, 65536>("tail: LUT global (L1), tile lds, scalar fma", out, glut);
        run_tail<0, true, 49152>("tail: LUT lds, tile lds, scalar fma", out, glut);
        run_tail<4, true, 49152>("tail: LUT global (L1), tile lds, scalar fma", out, glut);
        run_tail<3, true, 49152>("tail: no reads at all (pure VALU), scalar fma", out, glut);
        float *ztab;
        CHECK(hipMalloc(&ztab, 32 * 4));
        {
            float zh[32];
            for (int k = 0; k < 32; k++) zh[k] = -1.0f + k / 64.0f;
            CHECK(hipMemcpy(ztab, zh, sizeof(zh), hipMemcpyHostToDevice));
        }
        run_pipe<8, false, 2, 65536, 2>("pipelined 2px x 32pl, groups of 8 (w const)", out, ztab);
        run_pipe<4, false, 2, 65536, 2>("pipelined 2px x 32pl, groups of 4 (w const)", out, ztab);
        run_pipe<8, true, 2, 65536, 2>("pipelined 2px x 32pl, groups of 8 (general)", out, ztab);
        run_pipe<8, false, 2, 49152, 3>("pipelined 2px x 32pl, groups of 8 (w const)", out, ztab);
        run_pipe<8, true, 2, 49152, 3>("pipelined 2px x 32pl, groups of 8 (general)", out, ztab);
        run_pipe<8, false, 2, 32768, 4>("pipelined 2px x 32pl, groups of 8 (w const)", out, ztab);
        run_gather<0>("broadcast", out);
        run_gather<1>("consecutive per half wave", out);
        run_gather<2>("random", out);
        run_gather<3>("8 distinct", out);
        run_gather<4>("row step in half wave", out);
        run_gather<5>("2-byte stride, even start", out);
        run_gather<6>("2-byte stride, odd start", out);
        run_gather<7>("dword stride, misaligned by 2", out);
        if (argc > 1 && argv[1][0] == 'g') return 0;
        if (argc > 1) return 0;
    }
    test_mfma_semantics();
    run_mfma_rate<0>(out);
    run_mfma_rate<8>(out);
    run_mfma_rate<16>(out);
    run_mfma_rate<32>(out);
    run<0>("v_fma_f32", out, cyc);
    run<24>("v_add_f32", out, cyc);
    run<1>("v_and_b32", out, cyc);
    run<2>("v_lshrrev_b32", out, cyc);
    run<17>("v_lshlrev_b32", out, cyc);
    run<3>("v_add_u32", out, cyc);
    run<18>("v_sub_u32", out, cyc);
    run<4>("v_or_b32", out, cyc);
    run<19>("v_max_u32", out, cyc);
    run<5>("v_and_or_b32", out, cyc);
    run<6>("v_bfe_u32", out, cyc);
    run<7>("v_cndmask_b32", out, cyc);
    run<8>("v_med3_f32", out, cyc);
    run<9>("v_cmp_gt_f32", out, cyc);
    run<20>("v_cmp_lt_u32", out, cyc);
    run<10>("v_sad_u16", out, cyc);
    run<23>("v_sad_u8", out, cyc);
    run<11>("v_lshl_or_b32", out, cyc);
    run<12>("v_alignbit_b32", out, cyc);
    run<13>("v_mov_b32", out, cyc);
    run<14>("v_dot4_u32_u8", out, cyc);
    run<15>("v_perm_b32", out, cyc);
    run<16>("v_cvt_f32_ubyte0", out, cyc);
    run<21>("v_mad_u32_u24", out, cyc);
    run<22>("v_cvt_pk_u8_f32", out, cyc);
#define RUN_ALL(LB)                                                              \
    run_loop<0, LB>("VAR0 round-1 arithmetic (w const)", out);                   \
    run_loop<1, LB>("VAR1 fixed-point, VALU coords (w const)", out);             \
    run_loop<2, LB>("VAR2 fixed-point, s by MFMA (w const)", out);               \
    run_loop<3, LB>("VAR3 fixed-point, T by MFMA (w const)", out);               \
    run_loop<4, LB>("VAR4 fixed-point, VALU coords (general)", out);             \
    run_loop<5, LB>("VAR5 fixed-point, s by MFMA (general)", out);               \
    run_loop<6, LB>("VAR6 fixed-point, s + rcp guess by MFMA", out);
    RUN_ALL(65536)
    RUN_ALL(49152)
    return 0;
}
