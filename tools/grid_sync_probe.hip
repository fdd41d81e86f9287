// grid_sync_probe.hip -- what a grid-wide barrier costs on gfx950 (cooperative launch, cooperative_groups::grid_group::sync), per workgroup count:
// the price a persistent Farneback iteration kernel would pay per iteration instead of a launch.  Also: an empty kernel's
// launch-to-launch period on one stream, the floor the barrier has to beat.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/grid_sync_probe tools/grid_sync_probe.hip
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

__global__ __launch_bounds__(256) void syncs(int k, float *buf, int n)
{
    cg::grid_group grid = cg::this_grid();
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = buf[i % n];
    for (int it = 0; it < k; it++) {
        buf[i % n] = v + 1.0f;          // something every block writes and a neighbour reads after the barrier, as M does
        grid.sync();
        v = buf[(i + 256) % n];
    }
    buf[i % n] = v;
}

__global__ __launch_bounds__(256) void touch(float *buf, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    buf[i % n] = buf[(i + 256) % n] + 1.0f;
}

int main()
{
    float *buf;
    const int n = 1 << 20;
    CHECK(hipMalloc(&buf, n * sizeof(float)));
    CHECK(hipMemset(buf, 0, n * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int blocks : {12, 48, 120, 300, 600, 1024}) {
        int k = 200, nn = n;
        void *args[] = {&k, &buf, &nn};
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            CHECK(hipLaunchCooperativeKernel((const void *)syncs, dim3(blocks), dim3(256), args, 0, nullptr));
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // the same dependent step as separate launches
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < k; it++) touch<<<blocks, 256>>>(buf, n);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms2;
        CHECK(hipEventElapsedTime(&ms2, e0, e1));
        printf("%5d workgroups: grid.sync + dependent access %.2f us per step; the same step as a launch %.2f us\n", blocks, ms * 1e3 / k, ms2 * 1e3 / k);
    }
    return 0;
}
