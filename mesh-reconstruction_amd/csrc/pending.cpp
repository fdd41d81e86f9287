// pending.cpp -- ABI entry points of include/mvs.h whose kernels are not built yet.  They fail
// loudly (no CPU fallback); each moves to its own .hip file as its kernel lands.
#include "mvs_internal.hpp"

using namespace mvs;

extern "C" {

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
