// pending.cpp -- ABI entry points of include/mvs.h whose kernels are not built yet.  They fail
// loudly (no CPU fallback); each moves to its own .hip file as its kernel lands.
#include "mvs_internal.hpp"

using namespace mvs;

extern "C" {

int mvs_flow(mvs_ctx *ctx, const uint8_t *, const uint8_t *, int, float *)
{
    return fail(ctx, MVS_ESTATE, "mvs_flow: not built yet");
}

}  // extern "C"
