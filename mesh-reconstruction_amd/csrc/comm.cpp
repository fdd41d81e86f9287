// comm.cpp -- the sweep of ONE main view on several GPUs of one node behind the C ABI (mvs_comm_*, mvs_sweep_sharded).
//
// SURVEY.md section 8(b), multi-GPU row / north_star: the V side views are dealt to the GPUs, every GPU builds the packed volume
// of its views, and the volumes are summed over xGMI with RCCL.  Cells are integers (count << CS | sum), so the sum is exact and
// the depth map is bit-identical to the single-GPU one.  The exchange is a reduce-scatter by plane slices (rank r receives the
// summed cells of planes [r D/G, (r+1) D/G)), a partial depth selection per rank (mvs_sweep_argmin_partial), an all-gather of
// the 8-byte partial records and the merge in plane order (mvs_sweep_combine_partials): half the bytes of an all-reduce on the
// links.  When the plane count is not a multiple of the GPU count the volume is all-reduced in place instead.
// One host thread per GPU drives its context, as the C ABI asks ("calls on a context are serialised by the caller").
// RCCL is resolved with dlopen at mvs_comm_create: libmvs_hip.so itself has no link dependency on librccl, and a process that has
// already loaded one (PyTorch) shares it.
#include "mvs_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;

    bool load()
    {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) {
            error = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "unknown error");
            return false;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(handle, n);
            if (!p) error = std::string("librccl.so lacks ") + n;
            return p;
        };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        ReduceScatter = (decltype(ReduceScatter))sym("ncclReduceScatter");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        return CommInitAll && CommDestroy && ReduceScatter && AllGather && AllReduce && GetErrorString;
    }
};

}  // namespace

struct mvs_comm {
    int n = 0, W = 0, H = 0;
    std::vector<int> devices;
    std::vector<mvs_ctx *> ctx;
    std::vector<ncclComm_t> comms;
    std::vector<mvs::DevBuf> slice, part, parts;  // per rank: the plane slice it owns, its partial bests, everybody's partial bests
    Rccl rccl;
    char err[512] = {0};
};

using namespace mvs;

static char g_comm_err[512] = "no error";

static int comm_fail(mvs_comm *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    snprintf(c ? c->err : g_comm_err, 512, "%s", buf);
    return code;
}

extern "C" {

mvs_comm *mvs_comm_create(const int *devices, int n, int width, int height)
{
    if (!devices || n < 1 || n > 64) {
        comm_fail(nullptr, MVS_EINVAL, "mvs_comm_create: need 1..64 devices (n = %d)", n);
        return nullptr;
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) {
                comm_fail(nullptr, MVS_EINVAL, "mvs_comm_create: device %d listed twice (one rank per GPU)", devices[i]);
                return nullptr;
            }
    mvs_comm *c = new (std::nothrow) mvs_comm();
    if (!c) {
        comm_fail(nullptr, MVS_ENOMEM, "mvs_comm_create: out of host memory");
        return nullptr;
    }
    c->n = n;
    c->W = width;
    c->H = height;
    c->devices.assign(devices, devices + n);
    c->slice.resize(n);
    c->part.resize(n);
    c->parts.resize(n);
    for (int i = 0; i < n; i++) {
        mvs_ctx *x = mvs_create(devices[i], width, height);
        if (!x) {
            comm_fail(nullptr, MVS_EHIP, "mvs_comm_create: rank %d: %s", i, mvs_last_error(nullptr));
            mvs_comm_destroy(c);
            return nullptr;
        }
        c->ctx.push_back(x);
    }
    if (!c->rccl.load()) {
        comm_fail(nullptr, MVS_EHIP, "mvs_comm_create: %s", c->rccl.error.c_str());
        mvs_comm_destroy(c);
        return nullptr;
    }
    c->comms.assign(n, nullptr);
    const ncclResult_t r = c->rccl.CommInitAll(c->comms.data(), n, devices);
    if (r != ncclSuccess) {
        comm_fail(nullptr, MVS_EHIP, "mvs_comm_create: ncclCommInitAll failed: %s", c->rccl.GetErrorString(r));
        c->comms.clear();
        mvs_comm_destroy(c);
        return nullptr;
    }
    snprintf(c->err, sizeof(c->err), "no error");
    return c;
}

void mvs_comm_destroy(mvs_comm *c)
{
    if (!c) return;
    for (size_t i = 0; i < c->ctx.size(); i++) {
        (void)hipSetDevice(c->devices[i]);
        (void)mvs_synchronize(c->ctx[i]);
        for (DevBuf *b : {&c->slice[i], &c->part[i], &c->parts[i]})
            if (b->ptr) (void)hipFree(b->ptr);
    }
    for (ncclComm_t k : c->comms)
        if (k && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(k);
    for (mvs_ctx *x : c->ctx) mvs_destroy(x);
    delete c;
}

int mvs_comm_size(const mvs_comm *c) { return c ? c->n : MVS_EINVAL; }
mvs_ctx *mvs_comm_context(mvs_comm *c, int rank) { return (c && rank >= 0 && rank < c->n) ? c->ctx[rank] : nullptr; }
const char *mvs_comm_last_error(const mvs_comm *c) { return c ? c->err : g_comm_err; }

int mvs_sweep_sharded(mvs_comm *c, const float main_cam[16], const uint8_t *main_hw, int nviews, const float *side_cams,
                      const uint8_t *const *side_frames, int nplanes, float z_lo, float z_hi, float *depth_hw, float *cost_hw)
{
    if (!c) return MVS_EINVAL;
    if (!main_cam || !main_hw || !depth_hw || nviews < 0 || (nviews > 0 && (!side_cams || !side_frames)))
        return comm_fail(c, MVS_EINVAL, "mvs_sweep_sharded: null argument");
    if (nplanes < 1 || nplanes > 4096) return comm_fail(c, MVS_EINVAL, "mvs_sweep_sharded: nplanes=%d out of range 1..4096", nplanes);
    // Everything a rank could reject is checked HERE, before any rank enters a collective: a rank that returned early would leave the
    // others waiting in RCCL for ever.  (A HIP or RCCL failure in the middle of the exchange is not recoverable either way.)
    for (int v = 0; v < nviews; v++)
        if (!side_frames[v]) return comm_fail(c, MVS_EINVAL, "mvs_sweep_sharded: side_frames[%d] is null", v);
    const int view_limit = mvs_sweep_sampler(c->ctx[0]) == MVS_SAMPLER_FIXED ? 255 : 256;  // the SUMMED cells must hold every view's count
    if (nviews > view_limit) return comm_fail(c, MVS_EINVAL, "mvs_sweep_sharded: %d views, the sampler's cells hold at most %d", nviews, view_limit);
    for (int r = 1; r < c->n; r++)
        if (mvs_sweep_sampler(c->ctx[r]) != mvs_sweep_sampler(c->ctx[0]))
            return comm_fail(c, MVS_ESTATE, "mvs_sweep_sharded: rank %d uses another sampler than rank 0 (cells of different formats cannot be summed)", r);
    const int n = c->n;
    const size_t P = (size_t)c->W * c->H;
    // plane slices of equal size: reduce-scatter; otherwise (or with the test hook MVS_COMM_ALLREDUCE set) all-reduce in place
    const bool scatter = nplanes % n == 0 && getenv("MVS_COMM_ALLREDUCE") == nullptr;
    const int slice_planes = nplanes / n;
    std::vector<int> rc(n, MVS_OK);
    std::vector<std::string> msg(n);
    auto worker = [&](int r) {
        mvs_ctx *x = c->ctx[r];
        auto fail_here = [&](int code, const char *what, const char *detail) {
            rc[r] = code;
            msg[r] = std::string(what) + ": " + detail;
        };
        if (hipSetDevice(c->devices[r]) != hipSuccess) return fail_here(MVS_EHIP, "hipSetDevice", "failed");
        // view shard of this rank: a contiguous range, empty for ranks beyond the view count
        const int per = (nviews + n - 1) / n;
        const int v0 = std::min(r * per, nviews), vn = std::max(0, std::min(per, nviews - v0));
        int e;
        if ((e = mvs_sweep_set_main(x, main_cam, main_hw))) return fail_here(e, "mvs_sweep_set_main", mvs_last_error(x));
        if ((e = mvs_sweep_set_views(x, vn, side_cams + 16 * (size_t)v0, side_frames + v0))) return fail_here(e, "mvs_sweep_set_views", mvs_last_error(x));
        if ((e = mvs_sweep_set_planes(x, nplanes, z_lo, z_hi))) return fail_here(e, "mvs_sweep_set_planes", mvs_last_error(x));
        if ((e = mvs_sweep_run(x, 0, vn, MVS_SWEEP_VOLUME))) return fail_here(e, "mvs_sweep_run", mvs_last_error(x));
        size_t vol_bytes = 0;
        uint32_t *vol = (uint32_t *)mvs_sweep_volume_device(x, &vol_bytes);
        if (!vol) return fail_here(MVS_ESTATE, "mvs_sweep_volume_device", mvs_last_error(x));
        hipStream_t st = x->stream;
        ncclResult_t q;
        if (scatter) {
            if ((e = ensure(x, c->slice[r], (size_t)slice_planes * P * 4)) || (e = ensure(x, c->part[r], P * 8)) || (e = ensure(x, c->parts[r], (size_t)n * P * 8)))
                return fail_here(e, "device allocation", mvs_last_error(x));
            if ((q = c->rccl.ReduceScatter(vol, c->slice[r].ptr, (size_t)slice_planes * P, ncclUint32, ncclSum, c->comms[r], st)) != ncclSuccess)
                return fail_here(MVS_EHIP, "ncclReduceScatter", c->rccl.GetErrorString(q));
            if ((e = mvs_sweep_argmin_partial(x, c->slice[r].ptr, r * slice_planes, slice_planes, c->part[r].ptr)))
                return fail_here(e, "mvs_sweep_argmin_partial", mvs_last_error(x));
            if ((q = c->rccl.AllGather(c->part[r].ptr, c->parts[r].ptr, P, ncclUint64, c->comms[r], st)) != ncclSuccess)
                return fail_here(MVS_EHIP, "ncclAllGather", c->rccl.GetErrorString(q));
            if ((e = mvs_sweep_combine_partials(x, c->parts[r].ptr, n))) return fail_here(e, "mvs_sweep_combine_partials", mvs_last_error(x));
        } else {
            if ((q = c->rccl.AllReduce(vol, vol, (size_t)nplanes * P, ncclUint32, ncclSum, c->comms[r], st)) != ncclSuccess)
                return fail_here(MVS_EHIP, "ncclAllReduce", c->rccl.GetErrorString(q));
            if ((e = mvs_sweep_argmin(x))) return fail_here(e, "mvs_sweep_argmin", mvs_last_error(x));
        }
        if (r == 0) {
            if ((e = mvs_sweep_fetch(x, depth_hw, cost_hw, nullptr, nullptr))) return fail_here(e, "mvs_sweep_fetch", mvs_last_error(x));
        } else if ((e = mvs_synchronize(x))) {
            return fail_here(e, "mvs_synchronize", mvs_last_error(x));
        }
    };
    if (n == 1) {
        worker(0);
    } else {
        std::vector<std::thread> threads;
        for (int r = 0; r < n; r++) threads.emplace_back(worker, r);
        for (auto &t : threads) t.join();
    }
    for (int r = 0; r < n; r++)
        if (rc[r] != MVS_OK) return comm_fail(c, rc[r], "mvs_sweep_sharded: rank %d (device %d): %s", r, c->devices[r], msg[r].c_str());
    return MVS_OK;
}

}  // extern "C"
