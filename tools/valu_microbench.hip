// valu_microbench.hip -- issue-rate probe for the instructions the sweep kernel is built from.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_microbench.hip -o tools/valu_microbench
// Prints cycles per wave-instruction per SIMD (s_memtime) at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 16;  // independent chains

template <int OP>
__global__ void probe(float *out, unsigned long long *cyc, float seed)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float a[UNROLL];
    float b[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { a[i] = seed + i + threadIdx.x; b[i] = seed * 0.5f + i; }
    const float c0 = seed * 1.0001f, c1 = seed * 0.37f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[0]), "v"(*(double *)&b[2]));
            if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 4) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 6) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 8) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[0]));
            if (OP == 12) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[0]));
            if (OP == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 14) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 15) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
            if (OP == 16) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 17) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 18) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 19) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 20) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 21) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c0));
            if (OP == 22) asm volatile("v_cvt_pk_u16_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 23) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
            // f64 (the triangulation PCA and the Farneback box filter accumulate in double): operands are register pairs
            if (OP == 24) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[0]), "v"(*(double *)&b[2]));
            if (OP == 25) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[0]));
            if (OP == 26) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[0]));
            if (OP == 27) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*(double *)&a[i & ~1]) : "v"(c0));
            if (OP == 28) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(*(double *)&b[0]));
            // round 5: what sweep_exact_rect's row loop is made of, and what could replace parts of it
            if (OP == 29) { int sg; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(a[i])); asm volatile("" :: "s"(sg)); }
            if (OP == 30) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (OP == 31) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(c0));
            if (OP == 32) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 33) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
            if (OP == 34) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (OP == 35) { int sg; asm volatile("v_readlane_b32 %0, %1, 3\n\ts_nop 1\n\tv_fmac_f32 %2, %0, %3" : "=&s"(sg), "+v"(b[i]) , "+v"(a[i]) : "v"(c0)); }
            if (OP == 36) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]\n\tv_fma_mix_f32 %0, %0, %1, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\tv_fmac_f32 %0, %1, %2\n\tv_cvt_i32_f32 %0, %0\n\tv_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 4095];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// LDS gather probe: ds_read_b64 with per-lane addresses, consecutive quads
template <int WIDTH>
__global__ void probe_lds(float *out, unsigned long long *cyc, int stride)
{
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((float *)lds)[i] = (float)i;
    __syncthreads();
    int addr = ((threadIdx.x & 63) * stride * 8) & 32767 & ~15;
    float acc = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (WIDTH == 8) {
                float2 v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(i * 512));
                asm volatile("" ::"v"(v));
            } else {
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(i * 512));
                asm volatile("" ::"v"(v));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, float *out, unsigned long long *cyc, int lanes_per_instr)
{
    printf("%-22s", name);
    for (int wps : {1, 2, 3, 4, 8}) {  // waves per SIMD
        const int threads = 256;       // 4 waves per block -> one per SIMD
        const int blocks = 256 * wps;  // wps blocks per CU
        probe<OP><<<blocks, threads>>>(out, cyc, 1.25f);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        probe<OP><<<blocks, threads>>>(out, cyc, 1.25f);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks);
        CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto c : h) avg += c; avg /= blocks;
        const double instr_per_wave = (double)ITERS * UNROLL;
        // per SIMD: wps waves each issuing instr_per_wave in `avg` cycles (memtime ticks)
        printf("  w%d: %.2f tick/instr/SIMD (%.3f ms)", wps, avg / (instr_per_wave * wps), ms);
    }
    printf("\n");
}

int main()
{
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    printf("ticks are s_memtime units (constant 100 MHz on gfx9? compare ms). instr per wave = %d\n", ITERS * UNROLL);
    run<0>("v_fma_f32", out, cyc, 64);
    run<1>("v_pk_fma_f32", out, cyc, 64);
    run<2>("v_rcp_f32", out, cyc, 64);
    run<3>("v_fma_mix_f32", out, cyc, 64);
    run<4>("v_cvt_i32_f32", out, cyc, 64);
    run<5>("v_sad_u32", out, cyc, 64);
    run<6>("v_fract_f32", out, cyc, 64);
    run<7>("v_mul_i32_i24", out, cyc, 64);
    run<8>("v_lshl_add_u32", out, cyc, 64);
    run<9>("v_add_f32", out, cyc, 64);
    run<10>("v_mul_f32", out, cyc, 64);
    run<11>("v_pk_mul_f32", out, cyc, 64);
    run<12>("v_pk_add_f32", out, cyc, 64);
    run<13>("v_mad_u32_u24", out, cyc, 64);
    run<14>("v_pk_fma_f16", out, cyc, 64);
    run<15>("v_cvt_f32_ubyte1", out, cyc, 64);
    run<16>("v_mul_lo_u32", out, cyc, 64);
    run<17>("v_dot4_u32_u8", out, cyc, 64);
    run<18>("v_pk_mad_u16", out, cyc, 64);
    run<19>("v_perm_b32", out, cyc, 64);
    run<20>("v_med3_i32", out, cyc, 64);
    run<21>("v_cndmask_b32", out, cyc, 64);
    run<22>("v_cvt_pk_u16_u32", out, cyc, 64);
    run<23>("v_cvt_f32_f16", out, cyc, 64);
    run<24>("v_fma_f64", out, cyc, 64);
    run<25>("v_add_f64", out, cyc, 64);
    run<26>("v_mul_f64", out, cyc, 64);
    run<27>("v_cvt_f64_f32", out, cyc, 64);
    run<28>("v_cvt_f32_f64", out, cyc, 64);
    run<29>("v_readlane_b32", out, cyc, 64);
    run<30>("v_add_u32", out, cyc, 64);
    run<31>("v_cvt_pk_u8_f32", out, cyc, 64);
    run<32>("v_sad_u8", out, cyc, 64);
    run<33>("v_cvt_u32_f32", out, cyc, 64);
    run<34>("v_sad_u16", out, cyc, 64);
    run<35>("readlane+nop+fmac(s)", out, cyc, 64);
    run<36>("mix,mix,fmac,cvt,sad (5)", out, cyc, 64);
    for (int stride : {1, 2, 3}) {
        for (int wps : {1, 2, 4}) {
            probe_lds<8><<<256 * wps, 256>>>(out, cyc, stride);
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0));
            probe_lds<8><<<256 * wps, 256>>>(out, cyc, stride);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("ds_read_b64 stride %d w%d: %.3f ms for %d wave-instr per SIMD-wave -> %.2f ns/instr/CU\n", stride, wps, ms,
                   ITERS * UNROLL, ms * 1e6 / ((double)ITERS * UNROLL * wps * 4));
        }
    }
    // wall-clock calibration: v_fma chain time per instruction in ns
    return 0;
}
