// pending.cpp -- ABI entry points of include/mvs.h whose kernels are not built yet.  They fail
// loudly (no CPU fallback); each moves to its own .hip file as its kernel lands.
#include "mvs_internal.hpp"

using namespace mvs;

extern "C" {

int mvs_load_mesh(mvs_ctx *ctx, const float *, int, const int32_t *, int)
{
    return fail(ctx, MVS_ESTATE, "mvs_load_mesh: raster path not built yet");
}
int mvs_depth(mvs_ctx *ctx, const float *, float *) { return fail(ctx, MVS_ESTATE, "mvs_depth: raster path not built yet"); }
int mvs_projected(mvs_ctx *ctx, const float *, const uint8_t *, const float *, uint8_t *)
{
    return fail(ctx, MVS_ESTATE, "mvs_projected: raster path not built yet");
}
int mvs_mix_background(mvs_ctx *ctx, const uint8_t *, const uint8_t *, float *, uint8_t *)
{
    return fail(ctx, MVS_ESTATE, "mvs_mix_background: not built yet");
}
int mvs_compare(mvs_ctx *ctx, const uint8_t *, const uint8_t *, float *)
{
    return fail(ctx, MVS_ESTATE, "mvs_compare: not built yet");
}
int mvs_flow_remap(mvs_ctx *ctx, const float *, int, const uint8_t *, uint8_t *)
{
    return fail(ctx, MVS_ESTATE, "mvs_flow_remap: not built yet");
}
int mvs_flow(mvs_ctx *ctx, const uint8_t *, const uint8_t *, int, float *)
{
    return fail(ctx, MVS_ESTATE, "mvs_flow: not built yet");
}

}  // extern "C"
