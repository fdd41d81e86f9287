// comm.cpp -- the sweep of ONE main view on several GPUs of one node behind the C ABI (mvs_comm_*, mvs_sweep_sharded).
//
// SURVEY.md section 8(b), multi-GPU row / 8(e).  One host thread per GPU drives its context, as the C ABI asks ("calls on a context
// are serialised by the caller"); the threads are PERSISTENT (started by the first call that needs more than one rank, parked on a
// condition variable between calls, joined by mvs_comm_destroy) and a call hands them one job.  Three ways to split the work
// (mvs_comm_set_mode):
//   MVS_SHARD_ROWS (default)   every GPU holds all side views and sweeps a band of the main view's pixel rows with depth selection
//                              inside the kernel (rows are independent); NO data-path collective: the bands go straight
//                              from their GPUs into the caller's host maps (mvs_sweep_sharded) or, in the resident form, into rank 0's
//                              depth / cost maps by peer copies over xGMI (mvs_comm_run: 4 bytes per pixel and map).  The split that scales.
//   MVS_SHARD_VIEWS            the north_star's split: the V side views are dealt to the GPUs, every GPU builds the packed volume of
//                              its views, the volumes are summed over xGMI with RCCL -- an all-reduce per plane group on a second
//                              stream while the next group is being swept -- and depth is selected from the summed volume.
//   MVS_SHARD_VIEWS_SCATTER    the same split with half the bytes on the links and no overlap: reduce-scatter by plane slices, a partial
//                              selection per rank, all-gather of the 8-byte partials, merge in plane order (plane count a multiple
//                              of the GPU count; otherwise it runs as MVS_SHARD_VIEWS).
// Cells are integers (count << CS | sum): sums are exact, every mode returns the single-GPU depth map bit for bit.
// Two forms: mvs_sweep_sharded (host frames in, host maps out, one call) and the RESIDENT form -- mvs_comm_set_planes / _set_main /
// _set_views upload once (every rank keeps all views, so any mode can run on them), mvs_comm_run sweeps what is resident and leaves the
// maps on rank 0's GPU, mvs_comm_fetch downloads them: what a caller with a sequence, and bench.py --via-comm, time.
// Failure handling: every rank does its local work and its allocations first; the ranks then meet at a host-side barrier and look
// at a shared error flag BEFORE anybody enters a collective -- a rank that failed early makes all of them skip the exchange instead
// of leaving the others waiting in RCCL for ever.  A failure inside the exchange aborts every communicator (ncclCommAbort), which
// unblocks the other ranks; the communicator is unusable afterwards (calls return MVS_ESTATE).
// Abort is best effort: ncclCommAbort is called on every rank's communicator from the failing rank's thread while the other ranks'
// threads may be inside a collective on theirs -- that is what unblocks them; RCCL documents the call as safe from another thread, and the
// communicator is never used again.
// RCCL is resolved with dlopen at mvs_comm_create (libmvs_hip.so itself has no link dependency on librccl, and a process that has
// already loaded one -- PyTorch -- shares it), but only the two view-sharded modes need it: a default librccl that cannot be loaded leaves
// a communicator that serves MVS_SHARD_ROWS (no collective) and reports the loader's message when a views mode is selected; the RCCL
// communicators themselves (ncclCommInitAll) are created by the first sweep that exchanges anything.
// Test hooks (hooks.hpp: honoured only under MVS_TEST_HOOKS=1, read once by mvs_comm_create): MVS_RCCL_LIBRARY names another library
// file -- an explicit request, so a file that cannot be loaded fails mvs_comm_create (tests: a missing one, and the loopback stand-in of
// tests/loopback_rccl/ that lets n ranks share the one GPU of a test box); MVS_COMM_ALLOW_SAME_DEVICE=1 accepts a device listed more than
// once (RCCL itself refuses that; the loopback library does not); MVS_COMM_TEST_FAIL_RANK=r makes rank r fail in its local phase, as a
// device allocation would; MVS_COMM_ALLREDUCE=1 runs the scatter mode as the all-reduce pipeline.
#include "mvs_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;

    bool load(const std::string &override_name)
    {
        std::string last;
        for (const char *name : {override_name.empty() ? "librccl.so.1" : override_name.c_str(), override_name.empty() ? "librccl.so" : override_name.c_str()}) {
            (void)dlerror();
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
            const char *e = dlerror();  // ONE call: dlerror() clears the message it returns
            last = e ? e : "unknown error";
        }
        if (!handle) {
            error = std::string("cannot load librccl.so: ") + last;
            return false;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(handle, n);
            if (!p) error = std::string("librccl.so lacks ") + n;
            return p;
        };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
        ReduceScatter = (decltype(ReduceScatter))sym("ncclReduceScatter");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        return usable();
    }
    bool usable() const { return CommInitAll && CommDestroy && CommAbort && ReduceScatter && AllGather && AllReduce && GetErrorString; }
};

// all ranks meet here between their local work and the exchange (C++17: no std::barrier); reusable, lives with the communicator
struct HostBarrier {
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0, generation = 0, n = 1;
    void arrive_and_wait()
    {
        std::unique_lock<std::mutex> lock(m);
        const int gen = generation;
        if (++waiting == n) {
            waiting = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return gen != generation; });
        }
    }
};

// The rank threads: started once, parked between calls.  dispatch() hands every rank the same job (called with the rank) and returns when
// all of them are through it; rank 0's share runs on the CALLING thread (one wake-up less on the critical path, and n = 1 needs no thread).
struct RankThreads {
    std::vector<std::thread> threads;  // ranks 1 .. n-1
    std::mutex m;
    std::condition_variable start, done;
    const std::function<void(int)> *job = nullptr;
    long generation = 0;
    int remaining = 0;
    bool quit = false;

    void loop(int rank)
    {
        long seen = 0;
        for (;;) {
            const std::function<void(int)> *j;
            {
                std::unique_lock<std::mutex> lock(m);
                start.wait(lock, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
                j = job;
            }
            (*j)(rank);
            {
                std::lock_guard<std::mutex> lock(m);
                if (--remaining == 0) done.notify_one();
            }
        }
    }
    void dispatch(int n, const std::function<void(int)> &f)
    {
        if (n > 1) {
            std::lock_guard<std::mutex> lock(m);
            if (threads.empty())
                for (int r = 1; r < n; r++) threads.emplace_back(&RankThreads::loop, this, r);
            job = &f;
            remaining = n - 1;
            generation++;
        }
        if (n > 1) start.notify_all();
        f(0);
        if (n > 1) {
            std::unique_lock<std::mutex> lock(m);
            done.wait(lock, [&] { return remaining == 0; });
            job = nullptr;
        }
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lock(m);
            quit = true;
        }
        start.notify_all();
        for (auto &t : threads) t.join();
        threads.clear();
    }
};

// what one call asks of every rank (null / false members: leave what is resident alone)
struct Job {
    const char *who = "mvs_comm";
    bool set_planes = false;
    int nplanes = 0;
    float z_lo = 0.f, z_hi = 0.f;
    const float *main_cam = nullptr;
    const uint8_t *main_hw = nullptr;
    bool set_views = false, shard_upload = false;  // shard_upload: a views mode uploads only the rank's own views (the one-call form)
    int nviews = 0;
    const float *side_cams = nullptr;
    const uint8_t *const *side_frames = nullptr;
    bool run = false;
    unsigned run_flags = 0;                      // rows mode: MVS_SWEEP_VOLUME also materialises the band of the volume
    float *depth_hw = nullptr, *cost_hw = nullptr;  // run: the maps go to the caller's host memory; both null: they stay on rank 0's GPU
    int async_slot = -1;                         // rows, resident, mvs_comm_run_async: queue only; the maps land in rank 0's result pair of this slot
};

}  // namespace

struct mvs_comm {
    int n = 0, W = 0, H = 0;
    std::vector<int> devices;
    std::vector<mvs_ctx *> ctx;
    std::vector<ncclComm_t> comms;
    std::vector<mvs::DevBuf> slice, part, parts;  // per rank: the plane slice it owns, its partial bests, everybody's partial bests
    std::vector<hipStream_t> comm_stream;         // per rank: the collectives of MVS_SHARD_VIEWS run beside the sweep of the next plane group
    std::vector<std::vector<hipEvent_t>> events;  // per rank: plane group swept / plane group summed
    int mode = MVS_SHARD_ROWS;
    int plane_groups = 4;
    bool broken = false;  // a collective failed and the communicators were aborted
    std::mutex abort_once;  // (a member: communicators of one process do not share it)
    Rccl rccl;
    mvs::Hooks hooks;     // environment switches as read by mvs_comm_create (hooks.hpp)
    RankThreads ranks;
    HostBarrier meet;
    // what is resident on the ranks' GPUs (the resident form, and what mvs_sweep_sharded leaves behind)
    bool have_main = false, have_views = false, have_planes = false, have_result = false;
    int V = 0, D = 0;
    std::vector<int> res_v0, res_vn;  // per rank: the global view range its context holds
    // peer access between rank r's device and rank 0's (both directions enabled at mvs_comm_create): 1 = the band copies of the resident rows
    // mode travel GPU to GPU (xGMI, or the same device), 0 = the runtime stages them through host memory (still correct; reported)
    std::vector<int> peer;
    // mvs_comm_run_async (rows mode): up to two calls in flight.  Per rank and slot: the band's staging copy (so that the next sweep may overwrite
    // the context's maps while the band is still travelling), "band staged" / "band landed on rank 0" events, a gather stream beside the sweep's;
    // on rank 0's device one (depth, cost) pair of whole maps per slot -- what mvs_comm_wait makes the current result
    std::vector<hipStream_t> gather_stream;
    std::vector<std::array<mvs::DevBuf, 2>> stage;
    std::vector<std::array<hipEvent_t, 2>> staged, sent;
    std::vector<std::array<bool, 2>> sent_valid;
    mvs::DevBuf result[2][2];
    std::deque<int> pending;  // slots of the calls in flight, oldest first (-1: a call that completed synchronously, result in rank 0's context)
    long issued = 0;
    int result_slot = -1;     // >= 0: mvs_comm_fetch reads result[result_slot]; -1: rank 0's context
    // per call
    std::vector<int> rc;
    std::vector<std::string> msg;
    std::atomic<int> failed{0};
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
    const Hooks hooks = read_hooks();
    for (int i = 0; i < n && !hooks.comm_allow_same_device; i++)  // (test hook: n ranks on one GPU, with the loopback collective library)
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
    c->hooks = hooks;
    c->n = n;
    c->W = width;
    c->H = height;
    c->devices.assign(devices, devices + n);
    c->slice.resize(n);
    c->part.resize(n);
    c->parts.resize(n);
    c->comm_stream.assign(n, nullptr);
    c->events.resize(n);
    c->res_v0.assign(n, 0);
    c->res_vn.assign(n, 0);
    c->rc.assign(n, MVS_OK);
    c->msg.resize(n);
    c->meet.n = n;
    c->peer.assign(n, 1);
    c->gather_stream.assign(n, nullptr);
    c->stage.resize(n);
    c->staged.assign(n, std::array<hipEvent_t, 2>{{nullptr, nullptr}});
    c->sent.assign(n, std::array<hipEvent_t, 2>{{nullptr, nullptr}});
    c->sent_valid.assign(n, std::array<bool, 2>{{false, false}});
    // before anything touches a GPU (the error path of a missing library must not need one).  Only the view-sharded modes need RCCL:
    // a library named explicitly must load; the default one may be absent -- rows mode works without, a views mode then says why not
    if (!c->rccl.load(hooks.rccl_library) && !hooks.rccl_library.empty()) {
        comm_fail(nullptr, MVS_EHIP, "mvs_comm_create: %s", c->rccl.error.c_str());
        mvs_comm_destroy(c);
        return nullptr;
    }
    for (int i = 0; i < n; i++) {
        mvs_ctx *x = mvs_create(devices[i], width, height);
        if (!x) {
            comm_fail(nullptr, MVS_EHIP, "mvs_comm_create: rank %d: %s", i, mvs_last_error(nullptr));
            mvs_comm_destroy(c);
            return nullptr;
        }
        c->ctx.push_back(x);
    }
    snprintf(c->err, sizeof(c->err), "no error");
    // Peer access rank r <-> rank 0, both directions: the resident rows mode sends every band to rank 0 with hipMemcpyPeerAsync, which is a
    // GPU-to-GPU transfer over xGMI only when the two devices may address each other -- otherwise the runtime bounces the band through host
    // memory, silently.  Enabled here once ("already enabled" -- by this process's PyTorch, say -- is fine), the outcome kept per rank
    // (mvs_comm_peer_access) and named in mvs_comm_last_error right after creation; nothing is refused: a staged band is slow, not wrong.
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    std::string staged_ranks;
    for (int r = 1; r < n; r++) {
        if (devices[r] == devices[0]) continue;
        int can_r0 = 0, can_0r = 0;
        bool ok = hipDeviceCanAccessPeer(&can_r0, devices[r], devices[0]) == hipSuccess && hipDeviceCanAccessPeer(&can_0r, devices[0], devices[r]) == hipSuccess && can_r0 && can_0r;
        for (int dir = 0; dir < 2 && ok; dir++) {
            const int from = dir ? devices[0] : devices[r], to = dir ? devices[r] : devices[0];
            const hipError_t e = hipSetDevice(from) == hipSuccess ? hipDeviceEnablePeerAccess(to, 0) : hipErrorInvalidDevice;
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
            (void)hipGetLastError();  // ("already enabled" is sticky otherwise)
        }
        c->peer[r] = ok ? 1 : 0;
        if (!ok) staged_ranks += (staged_ranks.empty() ? "" : ", ") + std::to_string(r);
    }
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    if (!staged_ranks.empty())
        snprintf(c->err, sizeof(c->err), "note: no peer access between rank 0 (device %d) and rank(s) %s: the resident rows mode's band copies are staged through host memory",
                 devices[0], staged_ranks.c_str());
    return c;
}

// the RCCL communicators, created by the first sweep that exchanges anything (rows mode never does)
static int comm_init_rccl(mvs_comm *c, const char *who)
{
    if (!c->comms.empty()) return MVS_OK;
    if (!c->rccl.usable()) return comm_fail(c, MVS_EHIP, "%s: the view-sharded modes need RCCL: %s", who, c->rccl.error.c_str());
    c->comms.assign(c->n, nullptr);
    const ncclResult_t r = c->rccl.CommInitAll(c->comms.data(), c->n, c->devices.data());
    if (r != ncclSuccess) {
        c->comms.clear();
        return comm_fail(c, MVS_EHIP, "%s: ncclCommInitAll failed: %s", who, c->rccl.GetErrorString(r));
    }
    return MVS_OK;
}

void mvs_comm_destroy(mvs_comm *c)
{
    if (!c) return;
    c->ranks.stop();
    for (size_t i = 0; i < c->ctx.size(); i++) {
        (void)hipSetDevice(c->devices[i]);
        (void)mvs_synchronize(c->ctx[i]);
        for (DevBuf *b : {&c->slice[i], &c->part[i], &c->parts[i]})
            if (b->ptr) (void)hipFree(b->ptr);
        if (c->comm_stream[i]) {
            (void)hipStreamSynchronize(c->comm_stream[i]);
            (void)hipStreamDestroy(c->comm_stream[i]);
        }
        for (hipEvent_t e : c->events[i]) (void)hipEventDestroy(e);
        if (c->gather_stream[i]) {
            (void)hipStreamSynchronize(c->gather_stream[i]);
            (void)hipStreamDestroy(c->gather_stream[i]);
        }
        for (int k = 0; k < 2; k++) {
            if (c->stage[i][k].ptr) (void)hipFree(c->stage[i][k].ptr);
            if (c->staged[i][k]) (void)hipEventDestroy(c->staged[i][k]);
            if (c->sent[i][k]) (void)hipEventDestroy(c->sent[i][k]);
        }
        if (i == 0)
            for (auto &pair : c->result)
                for (DevBuf &b : pair)
                    if (b.ptr) (void)hipFree(b.ptr);
    }
    for (ncclComm_t k : c->comms)
        if (k && !c->broken && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(k);  // (aborted communicators are already gone)
    for (mvs_ctx *x : c->ctx) mvs_destroy(x);
    delete c;
}

int mvs_comm_size(const mvs_comm *c) { return c ? c->n : MVS_EINVAL; }
int mvs_comm_peer_access(const mvs_comm *c, int rank) { return (c && rank >= 0 && rank < c->n) ? c->peer[rank] : MVS_EINVAL; }
int mvs_comm_device(const mvs_comm *c, int rank) { return (c && rank >= 0 && rank < c->n) ? c->devices[rank] : MVS_EINVAL; }
mvs_ctx *mvs_comm_context(mvs_comm *c, int rank) { return (c && rank >= 0 && rank < c->n) ? c->ctx[rank] : nullptr; }
const char *mvs_comm_last_error(const mvs_comm *c) { return c ? c->err : g_comm_err; }

int mvs_comm_set_mode(mvs_comm *c, int mode)
{
    if (!c) return MVS_EINVAL;
    if (mode != MVS_SHARD_ROWS && mode != MVS_SHARD_VIEWS && mode != MVS_SHARD_VIEWS_SCATTER) return comm_fail(c, MVS_EINVAL, "mvs_comm_set_mode: unknown mode %d", mode);
    if (mode != MVS_SHARD_ROWS && !c->rccl.usable()) return comm_fail(c, MVS_EHIP, "mvs_comm_set_mode: the view-sharded modes need RCCL: %s", c->rccl.error.c_str());
    c->mode = mode;
    return MVS_OK;
}

int mvs_comm_mode(const mvs_comm *c) { return c ? c->mode : MVS_EINVAL; }

int mvs_comm_set_plane_groups(mvs_comm *c, int groups)
{
    if (!c) return MVS_EINVAL;
    if (groups < 1 || groups > 64) return comm_fail(c, MVS_EINVAL, "mvs_comm_set_plane_groups: %d out of range 1..64", groups);
    c->plane_groups = groups;
    return MVS_OK;
}

}  // extern "C"

// One job on every rank.  The caller has checked the arguments; this function owns the sharding arithmetic, the two phases and the
// failure protocol described at the top of the file.
static int comm_execute(mvs_comm *c, const Job &job)
{
    const int n = c->n;
    const int W = c->W, H = c->H;
    const size_t P = (size_t)W * H;
    const bool rows = c->mode == MVS_SHARD_ROWS;
    const int nviews = job.set_views ? job.nviews : c->V;
    const int nplanes = job.set_planes ? job.nplanes : c->D;
    const bool exchange = job.run && !rows;                                       // RCCL collectives
    const bool queued = job.run && rows && job.async_slot >= 0;                   // mvs_comm_run_async: launches only, nobody waits for a GPU
    const bool gather = job.run && rows && !job.depth_hw && !job.cost_hw && (n > 1 || queued);  // rows, resident: the bands travel to rank 0 by peer copies
    const int slot = job.async_slot;
    if (exchange) {
        const int e = comm_init_rccl(c, job.who);
        if (e) return e;
    }
    // plane slices of equal size: reduce-scatter; otherwise (or with the test hook MVS_COMM_ALLREDUCE set) the all-reduce pipeline
    const bool scatter = c->mode == MVS_SHARD_VIEWS_SCATTER && nplanes > 0 && nplanes % n == 0 && !c->hooks.comm_allreduce;
    const int slice_planes = nplanes / n;
    // row bands on the sweep's tile-row granularity, equal but for the last
    const int gran = mvs_sweep_row_granularity_of(c->ctx[0]);
    const int band = ((H + gran - 1) / gran + n - 1) / n * gran;
    // plane groups of the all-reduce pipeline, on the sweep's plane granularity
    std::vector<std::pair<int, int>> groups;
    if (exchange && !scatter) {
        const int pg = mvs_sweep_plane_granularity();
        const int per = std::max(pg, ((nplanes + c->plane_groups - 1) / c->plane_groups + pg - 1) / pg * pg);
        for (int first = 0; first < nplanes; first += per) groups.emplace_back(first, std::min(per, nplanes - first));
    }
    std::fill(c->rc.begin(), c->rc.end(), MVS_OK);
    for (auto &m : c->msg) m.clear();
    c->failed.store(0);
    auto abort_all = [&]() {  // a collective failed on this rank: unblock everybody else (RCCL tolerates a second abort of a dead communicator poorly, so once)
        std::lock_guard<std::mutex> lock(c->abort_once);
        if (c->broken) return;
        c->broken = true;
        for (ncclComm_t k : c->comms)
            if (k) (void)c->rccl.CommAbort(k);
    };
    const std::function<void(int)> worker = [&](int r) {
        mvs_ctx *x = c->ctx[r];
        auto fail_here = [&](int code, const char *what, const char *detail) {
            c->rc[r] = code;
            c->msg[r] = std::string(what) + ": " + detail;
            c->failed.store(1);
        };
        uint32_t *vol = nullptr;
        hipStream_t st = nullptr;
        const int per = (nviews + n - 1) / n;  // view shard of this rank: a contiguous range, empty for ranks beyond the view count
        const int v0 = rows ? 0 : std::min(r * per, nviews), vn = rows ? nviews : std::max(0, std::min(per, nviews - v0));
        const int r0 = std::min(r * band, H), rn = std::max(0, std::min(band, H - r0));
        int lv0 = v0;  // first view of the shard as the rank's context numbers its resident views
        // ---- phase 1: everything local (inputs, allocations; in rows mode the whole sweep) ----
        [&]() {
            if (hipSetDevice(c->devices[r]) != hipSuccess) return fail_here(MVS_EHIP, "hipSetDevice", "failed");
            if (r == c->hooks.comm_fail_rank && (job.run || job.set_views)) return fail_here(MVS_ENOMEM, "test hook", "MVS_COMM_TEST_FAIL_RANK names this rank");
            int e;
            st = x->stream;
            // planes before views: mvs_sweep_set_views plans with the planes it finds (one planner pass per call, not two)
            if (job.set_planes && (e = sweep_set_planes_impl(x, job.nplanes, job.z_lo, job.z_hi, false))) return fail_here(e, "mvs_sweep_set_planes", mvs_last_error(x));
            if (job.main_hw && (e = sweep_set_main_impl(x, job.main_cam, job.main_hw, false))) return fail_here(e, "mvs_sweep_set_main", mvs_last_error(x));
            if (job.set_views) {
                const int u0 = job.shard_upload ? v0 : 0, un = job.shard_upload ? vn : nviews;
                if ((e = sweep_set_views_impl(x, un, job.side_cams + 16 * (size_t)u0, job.side_frames + u0, false))) return fail_here(e, "mvs_sweep_set_views", mvs_last_error(x));
                c->res_v0[r] = u0;
                c->res_vn[r] = un;
            }
            if (!job.run) {
                if ((e = mvs_synchronize(x))) return fail_here(e, "mvs_synchronize", mvs_last_error(x));  // the caller's buffers are not retained
                return;
            }
            lv0 = v0 - c->res_v0[r];
            if (lv0 < 0 || lv0 + vn > c->res_vn[r])
                return fail_here(MVS_ESTATE, job.who, "this rank does not hold the views the mode needs (uploaded by mvs_sweep_sharded under another mode): call mvs_comm_set_views");
            if (rows) {
                if (gather && (x->depth.bytes < P * 4 || x->cost.bytes < P * 4)) {
                    // first resident run: the maps are allocated (and, under the poison hook, filled) HERE and the fill waited for, so that nothing
                    // whole-map is ever queued on rank 0's stream behind the meet -- a later fill would land on top of the other ranks' bands
                    if ((e = ensure(x, x->depth, P * 4)) || (e = ensure(x, x->cost, P * 4)) || (e = mvs_synchronize(x))) return fail_here(e, "device allocation", mvs_last_error(x));
                }
                if (queued) {
                    if (!c->gather_stream[r] && hipStreamCreateWithFlags(&c->gather_stream[r], hipStreamNonBlocking) != hipSuccess) return fail_here(MVS_EHIP, "hipStreamCreate", "failed");
                    for (hipEvent_t *ev : {&c->staged[r][slot], &c->sent[r][slot]})
                        if (!*ev && hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) return fail_here(MVS_EHIP, "hipEventCreate", "failed");
                    if (r == 0)
                        for (DevBuf &b : c->result[slot])
                            if (b.bytes < P * 4) {
                                if ((e = ensure(x, b, P * 4)) || (e = mvs_synchronize(x))) return fail_here(e, "device allocation", mvs_last_error(x));
                            }
                    if (r > 0 && rn > 0 && c->stage[r][slot].bytes < (size_t)rn * W * 8) {
                        if ((e = ensure(x, c->stage[r][slot], (size_t)rn * W * 8)) || (e = mvs_synchronize(x))) return fail_here(e, "device allocation", mvs_last_error(x));
                    }
                    // the slot's staging buffer is free again when the band of the call two back has landed (a wait on the GPU, not the host)
                    if (c->sent_valid[r][slot] && r > 0 && hipStreamWaitEvent(st, c->sent[r][slot], 0) != hipSuccess) return fail_here(MVS_EHIP, "hipStreamWaitEvent", "failed");
                }
                if (rn > 0) {
                    if ((e = mvs_sweep_run_rows(x, 0, nviews, r0, rn, MVS_SWEEP_FUSED_ARGMIN | (job.run_flags & MVS_SWEEP_VOLUME)))) return fail_here(e, "mvs_sweep_run_rows", mvs_last_error(x));
                    if (queued && r > 0) {
                        // the band leaves the context's maps at once (a device copy of 8 bytes per band pixel): the next call's sweep may overwrite them
                        float *sd = (float *)c->stage[r][slot].ptr;
                        const size_t off = (size_t)r0 * W, cells = (size_t)rn * W;
                        if (hipMemcpyAsync(sd, (const float *)x->depth.ptr + off, cells * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                            hipMemcpyAsync(sd + cells, (const float *)x->cost.ptr + off, cells * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                            hipEventRecord(c->staged[r][slot], st) != hipSuccess)
                            return fail_here(MVS_EHIP, "hipMemcpyAsync", "band to its staging buffer");
                    }
                    // the one-call form: the band goes straight into the caller's maps: 4 bytes per pixel and map, no collective
                    if (job.depth_hw && hipMemcpyAsync(job.depth_hw + (size_t)r0 * W, (const float *)x->depth.ptr + (size_t)r0 * W, (size_t)rn * W * 4, hipMemcpyDeviceToHost, st) != hipSuccess)
                        return fail_here(MVS_EHIP, "hipMemcpyAsync", "depth band");
                    if (job.cost_hw && hipMemcpyAsync(job.cost_hw + (size_t)r0 * W, (const float *)x->cost.ptr + (size_t)r0 * W, (size_t)rn * W * 4, hipMemcpyDeviceToHost, st) != hipSuccess)
                        return fail_here(MVS_EHIP, "hipMemcpyAsync", "cost band");
                }   // (rank 0 always has rows: its maps exist when the others copy their bands into them)
                if (!gather && (e = mvs_synchronize(x))) return fail_here(e, "mvs_synchronize", mvs_last_error(x));
                return;
            }
            // views: the first sweep launch allocates the volume and the outputs; the buffers of the exchange follow
            if (scatter) {
                if ((e = mvs_sweep_run(x, lv0, vn, MVS_SWEEP_VOLUME))) return fail_here(e, "mvs_sweep_run", mvs_last_error(x));
                if ((e = ensure(x, c->slice[r], (size_t)slice_planes * P * 4)) || (e = ensure(x, c->part[r], P * 8)) || (e = ensure(x, c->parts[r], (size_t)n * P * 8)))
                    return fail_here(e, "device allocation", mvs_last_error(x));
            } else {
                if (!c->comm_stream[r] && hipStreamCreateWithFlags(&c->comm_stream[r], hipStreamNonBlocking) != hipSuccess) return fail_here(MVS_EHIP, "hipStreamCreate", "failed");
                while (c->events[r].size() < 2 * groups.size()) {
                    hipEvent_t ev;
                    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail_here(MVS_EHIP, "hipEventCreate", "failed");
                    c->events[r].push_back(ev);
                }
                if ((e = mvs_sweep_run_planes(x, lv0, vn, groups[0].first, groups[0].second, MVS_SWEEP_VOLUME))) return fail_here(e, "mvs_sweep_run_planes", mvs_last_error(x));
            }
            size_t vol_bytes = 0;
            vol = (uint32_t *)mvs_sweep_volume_device(x, &vol_bytes);
            if (!vol) return fail_here(MVS_ESTATE, "mvs_sweep_volume_device", mvs_last_error(x));
        }();
        if (!exchange && !gather) return;
        // ---- every rank is through its local work: does anybody have to give up? ----
        c->meet.arrive_and_wait();
        if (c->failed.load()) {
            if (c->rc[r] == MVS_OK) (void)mvs_synchronize(x);
            return;
        }
        int e;
        if (queued) {
            // mvs_comm_run_async: rank 0 copies its own band into the slot's result pair behind its sweep; every other rank sends its staged band
            // there on its gather stream -- beside the NEXT call's sweep on the rank's main stream.  Nobody waits: mvs_comm_wait does.
            float *rd = (float *)c->result[slot][0].ptr, *rc_ = (float *)c->result[slot][1].ptr;
            const size_t off = (size_t)r0 * W, cells = (size_t)rn * W;
            if (rn > 0 && r == 0) {
                if (hipMemcpyAsync(rd + off, (const float *)x->depth.ptr + off, cells * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                    hipMemcpyAsync(rc_ + off, (const float *)x->cost.ptr + off, cells * 4, hipMemcpyDeviceToDevice, st) != hipSuccess || hipEventRecord(c->sent[r][slot], st) != hipSuccess)
                    fail_here(MVS_EHIP, "hipMemcpyAsync", "rank 0's band to the result maps");
                else
                    c->sent_valid[r][slot] = true;
            } else if (rn > 0) {
                hipStream_t gs = c->gather_stream[r];
                const float *sd = (const float *)c->stage[r][slot].ptr;
                if (hipStreamWaitEvent(gs, c->staged[r][slot], 0) != hipSuccess ||
                    hipMemcpyPeerAsync(rd + off, c->devices[0], sd, c->devices[r], cells * 4, gs) != hipSuccess ||
                    hipMemcpyPeerAsync(rc_ + off, c->devices[0], sd + cells, c->devices[r], cells * 4, gs) != hipSuccess || hipEventRecord(c->sent[r][slot], gs) != hipSuccess)
                    fail_here(MVS_EHIP, "hipMemcpyPeerAsync", "band to rank 0");
                else
                    c->sent_valid[r][slot] = true;
            }
            return;
        }
        if (gather) {
            // rows, resident: rank 0's maps exist (allocated and filled before the meet) and every other rank copies its band into them over xGMI -- a
            // peer copy on the rank's own stream behind its sweep; 4 bytes per pixel and map in total, no collective, no host memory
            if (r > 0 && rn > 0) {
                mvs_ctx *x0 = c->ctx[0];
                const size_t off = (size_t)r0 * W, bytes = (size_t)rn * W * 4;
                if (hipMemcpyPeerAsync((float *)x0->depth.ptr + off, c->devices[0], (const float *)x->depth.ptr + off, c->devices[r], bytes, st) != hipSuccess ||
                    hipMemcpyPeerAsync((float *)x0->cost.ptr + off, c->devices[0], (const float *)x->cost.ptr + off, c->devices[r], bytes, st) != hipSuccess)
                    fail_here(MVS_EHIP, "hipMemcpyPeerAsync", "band to rank 0");
            }
            if ((e = mvs_synchronize(x))) fail_here(e, "mvs_synchronize", mvs_last_error(x));
            return;
        }
        // ---- phase 2: the exchange ----
        auto collective_failed = [&](const char *what, ncclResult_t q) {
            fail_here(MVS_EHIP, what, c->rccl.GetErrorString(q));
            abort_all();
        };
        ncclResult_t q;
        if (scatter) {
            if ((q = c->rccl.ReduceScatter(vol, c->slice[r].ptr, (size_t)slice_planes * P, ncclUint32, ncclSum, c->comms[r], st)) != ncclSuccess) return collective_failed("ncclReduceScatter", q);
            if ((e = mvs_sweep_argmin_partial(x, c->slice[r].ptr, r * slice_planes, slice_planes, c->part[r].ptr))) {
                fail_here(e, "mvs_sweep_argmin_partial", mvs_last_error(x));
                return abort_all();
            }
            if ((q = c->rccl.AllGather(c->part[r].ptr, c->parts[r].ptr, P, ncclUint64, c->comms[r], st)) != ncclSuccess) return collective_failed("ncclAllGather", q);
            if ((e = mvs_sweep_combine_partials(x, c->parts[r].ptr, n))) {
                fail_here(e, "mvs_sweep_combine_partials", mvs_last_error(x));
                return abort_all();
            }
        } else {
            // plane group g is summed over xGMI on the communication stream while group g + 1 is being swept on the context's stream
            hipStream_t cs = c->comm_stream[r];
            for (size_t g = 0; g < groups.size(); g++) {
                if (g > 0 && (e = mvs_sweep_run_planes(x, lv0, vn, groups[g].first, groups[g].second, MVS_SWEEP_VOLUME))) {
                    fail_here(e, "mvs_sweep_run_planes", mvs_last_error(x));
                    return abort_all();
                }
                hipEvent_t swept = c->events[r][2 * g], summed = c->events[r][2 * g + 1];
                if (hipEventRecord(swept, st) != hipSuccess || hipStreamWaitEvent(cs, swept, 0) != hipSuccess) {
                    fail_here(MVS_EHIP, "hipEventRecord / hipStreamWaitEvent", "failed");
                    return abort_all();
                }
                uint32_t *grp = vol + (size_t)groups[g].first * P;
                if ((q = c->rccl.AllReduce(grp, grp, (size_t)groups[g].second * P, ncclUint32, ncclSum, c->comms[r], cs)) != ncclSuccess) return collective_failed("ncclAllReduce", q);
                if (hipEventRecord(summed, cs) != hipSuccess) {
                    fail_here(MVS_EHIP, "hipEventRecord", "failed");
                    return abort_all();
                }
            }
            for (size_t g = 0; g < groups.size(); g++)
                if (hipStreamWaitEvent(st, c->events[r][2 * g + 1], 0) != hipSuccess) {
                    fail_here(MVS_EHIP, "hipStreamWaitEvent", "failed");
                    return abort_all();
                }
            if ((e = mvs_sweep_argmin(x))) {
                fail_here(e, "mvs_sweep_argmin", mvs_last_error(x));
                return abort_all();
            }
        }
        if (r == 0 && (job.depth_hw || job.cost_hw)) {
            if ((e = mvs_sweep_fetch(x, job.depth_hw, job.cost_hw, nullptr, nullptr))) fail_here(e, "mvs_sweep_fetch", mvs_last_error(x));
        } else if ((e = mvs_synchronize(x))) {
            fail_here(e, "mvs_synchronize", mvs_last_error(x));
        }
    };
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);   // rank 0's share runs on the calling thread and selects rank 0's device: put the caller's back afterwards
    c->ranks.dispatch(n, worker);
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    // new inputs: the maps on rank 0 belong to a view that is no longer resident -- a fetch before the next run must not return them as current
    if (job.set_planes || job.main_hw || job.set_views) c->have_result = false;
    // what the ranks hold now (a failed upload leaves the communicator without that input, like a context)
    bool ok = true;
    for (int r = 0; r < n; r++) ok = ok && c->rc[r] == MVS_OK;
    if (job.set_planes) {
        c->have_planes = ok;
        c->D = job.nplanes;
    }
    if (job.main_hw) {
        c->have_main = ok;
        c->have_views = false;  // the view matrices depend on the main camera (as mvs_sweep_set_main)
    }
    if (job.set_views) {
        c->have_views = ok;
        c->V = job.nviews;
    }
    if (job.run && !queued) {
        c->have_result = ok;
        c->result_slot = -1;
    }
    for (int r = 0; r < n; r++)
        if (c->rc[r] != MVS_OK) return comm_fail(c, c->rc[r], "%s: rank %d (device %d): %s", job.who, r, c->devices[r], c->msg[r].c_str());
    return MVS_OK;
}

static int comm_check_views(mvs_comm *c, const char *who, int nviews)
{
    const int view_limit = mvs_sweep_sampler(c->ctx[0]) == MVS_SAMPLER_FIXED ? 255 : 256;  // the SUMMED cells must hold every view's count
    if (nviews > view_limit) return comm_fail(c, MVS_EINVAL, "%s: %d views, the sampler's cells hold at most %d", who, nviews, view_limit);
    for (int r = 1; r < c->n; r++)
        if (mvs_sweep_sampler(c->ctx[r]) != mvs_sweep_sampler(c->ctx[0]))
            return comm_fail(c, MVS_ESTATE, "%s: rank %d uses another sampler than rank 0 (cells of different formats cannot be summed)", who, r);
    return MVS_OK;
}

extern "C" {

int mvs_sweep_sharded(mvs_comm *c, const float main_cam[16], const uint8_t *main_hw, int nviews, const float *side_cams,
                      const uint8_t *const *side_frames, int nplanes, float z_lo, float z_hi, float *depth_hw, float *cost_hw)
{
    if (!c) return MVS_EINVAL;
    if (c->broken) return comm_fail(c, MVS_ESTATE, "mvs_sweep_sharded: the communicators were aborted after a failed collective; create a new mvs_comm");
    if (!c->pending.empty()) return comm_fail(c, MVS_ESTATE, "mvs_sweep_sharded: %d call(s) of mvs_comm_run_async in flight: mvs_comm_wait first", (int)c->pending.size());
    if (!main_cam || !main_hw || !depth_hw || nviews < 0 || (nviews > 0 && (!side_cams || !side_frames)))
        return comm_fail(c, MVS_EINVAL, "mvs_sweep_sharded: null argument");
    if (nplanes < 1 || nplanes > 4096) return comm_fail(c, MVS_EINVAL, "mvs_sweep_sharded: nplanes=%d out of range 1..4096", nplanes);
    // everything a rank could reject for the arguments' sake is checked here, once
    for (int v = 0; v < nviews; v++)
        if (!side_frames[v]) return comm_fail(c, MVS_EINVAL, "mvs_sweep_sharded: side_frames[%d] is null", v);
    int e = comm_check_views(c, "mvs_sweep_sharded", nviews);
    if (e) return e;
    Job job;
    job.who = "mvs_sweep_sharded";
    job.set_planes = true;
    job.nplanes = nplanes;
    job.z_lo = z_lo;
    job.z_hi = z_hi;
    job.main_cam = main_cam;
    job.main_hw = main_hw;
    job.set_views = true;
    job.shard_upload = c->mode != MVS_SHARD_ROWS;  // a views mode: every rank uploads its own views only
    job.nviews = nviews;
    job.side_cams = side_cams;
    job.side_frames = side_frames;
    job.run = true;
    job.depth_hw = depth_hw;
    job.cost_hw = cost_hw;
    return comm_execute(c, job);
}

int mvs_comm_set_planes(mvs_comm *c, int nplanes, float z_lo, float z_hi)
{
    if (!c) return MVS_EINVAL;
    if (c->broken) return comm_fail(c, MVS_ESTATE, "mvs_comm_set_planes: the communicators were aborted after a failed collective; create a new mvs_comm");
    if (!c->pending.empty()) return comm_fail(c, MVS_ESTATE, "mvs_comm_set_planes: %d call(s) of mvs_comm_run_async in flight: mvs_comm_wait first", (int)c->pending.size());
    if (nplanes < 1 || nplanes > 4096) return comm_fail(c, MVS_EINVAL, "mvs_comm_set_planes: nplanes=%d out of range 1..4096", nplanes);
    Job job;
    job.who = "mvs_comm_set_planes";
    job.set_planes = true;
    job.nplanes = nplanes;
    job.z_lo = z_lo;
    job.z_hi = z_hi;
    return comm_execute(c, job);
}

int mvs_comm_set_main(mvs_comm *c, const float main_cam[16], const uint8_t *main_hw)
{
    if (!c) return MVS_EINVAL;
    if (c->broken) return comm_fail(c, MVS_ESTATE, "mvs_comm_set_main: the communicators were aborted after a failed collective; create a new mvs_comm");
    if (!c->pending.empty()) return comm_fail(c, MVS_ESTATE, "mvs_comm_set_main: %d call(s) of mvs_comm_run_async in flight: mvs_comm_wait first", (int)c->pending.size());
    if (!main_cam || !main_hw) return comm_fail(c, MVS_EINVAL, "mvs_comm_set_main: null argument");
    Job job;
    job.who = "mvs_comm_set_main";
    job.main_cam = main_cam;
    job.main_hw = main_hw;
    return comm_execute(c, job);
}

int mvs_comm_set_views(mvs_comm *c, int nviews, const float *side_cams, const uint8_t *const *side_frames)
{
    if (!c) return MVS_EINVAL;
    if (c->broken) return comm_fail(c, MVS_ESTATE, "mvs_comm_set_views: the communicators were aborted after a failed collective; create a new mvs_comm");
    if (!c->pending.empty()) return comm_fail(c, MVS_ESTATE, "mvs_comm_set_views: %d call(s) of mvs_comm_run_async in flight: mvs_comm_wait first", (int)c->pending.size());
    if (nviews < 0 || (nviews > 0 && (!side_cams || !side_frames))) return comm_fail(c, MVS_EINVAL, "mvs_comm_set_views: null argument");
    if (!c->have_main) return comm_fail(c, MVS_ESTATE, "mvs_comm_set_views: call mvs_comm_set_main first");
    for (int v = 0; v < nviews; v++)
        if (!side_frames[v]) return comm_fail(c, MVS_EINVAL, "mvs_comm_set_views: side_frames[%d] is null", v);
    int e = comm_check_views(c, "mvs_comm_set_views", nviews);
    if (e) return e;
    Job job;
    job.who = "mvs_comm_set_views";
    job.set_views = true;
    job.nviews = nviews;
    job.side_cams = side_cams;
    job.side_frames = side_frames;
    return comm_execute(c, job);
}

int mvs_comm_run(mvs_comm *c, unsigned flags)
{
    if (!c) return MVS_EINVAL;
    if (c->broken) return comm_fail(c, MVS_ESTATE, "mvs_comm_run: the communicators were aborted after a failed collective; create a new mvs_comm");
    if (!c->pending.empty()) return comm_fail(c, MVS_ESTATE, "mvs_comm_run: %d call(s) of mvs_comm_run_async in flight: mvs_comm_wait first", (int)c->pending.size());
    if (!c->have_main || !c->have_views || !c->have_planes)
        return comm_fail(c, MVS_ESTATE, "mvs_comm_run: set planes, main view and side views first (mvs_comm_set_planes / _set_main / _set_views)");
    if (flags & ~(unsigned)MVS_SWEEP_VOLUME) return comm_fail(c, MVS_EINVAL, "mvs_comm_run: flags may only carry MVS_SWEEP_VOLUME");
    int e = comm_check_views(c, "mvs_comm_run", c->V);
    if (e) return e;
    Job job;
    job.who = "mvs_comm_run";
    job.run = true;
    job.run_flags = flags;
    return comm_execute(c, job);
}

int mvs_comm_run_async(mvs_comm *c, unsigned flags)
{
    if (!c) return MVS_EINVAL;
    if (c->broken) return comm_fail(c, MVS_ESTATE, "mvs_comm_run_async: the communicators were aborted after a failed collective; create a new mvs_comm");
    if (!c->have_main || !c->have_views || !c->have_planes)
        return comm_fail(c, MVS_ESTATE, "mvs_comm_run_async: set planes, main view and side views first (mvs_comm_set_planes / _set_main / _set_views)");
    if (flags) return comm_fail(c, MVS_EINVAL, "mvs_comm_run_async: flags must be 0");
    if (c->pending.size() >= 2) return comm_fail(c, MVS_ESTATE, "mvs_comm_run_async: two calls are in flight already: mvs_comm_wait for the older one first");
    if (!c->pending.empty() && (c->mode != MVS_SHARD_ROWS || c->pending.front() < 0))
        return comm_fail(c, MVS_ESTATE, "mvs_comm_run_async: only MVS_SHARD_ROWS keeps two calls in flight (a view-sharded call completes inside mvs_comm_run_async): mvs_comm_wait first");
    int e = comm_check_views(c, "mvs_comm_run_async", c->V);
    if (e) return e;
    Job job;
    job.who = "mvs_comm_run_async";
    job.run = true;
    if (c->mode == MVS_SHARD_ROWS) {
        job.async_slot = (int)(c->issued & 1);
        if ((e = comm_execute(c, job))) return e;
        c->pending.push_back(job.async_slot);
    } else {
        // the view-sharded modes end in collectives every rank thread has to drive: they run to completion here (documented: include/mvs.h)
        if ((e = comm_execute(c, job))) return e;
        c->have_result = false;
        c->pending.push_back(-1);
    }
    c->issued++;
    return MVS_OK;
}

int mvs_comm_wait(mvs_comm *c)
{
    if (!c) return MVS_EINVAL;
    if (c->pending.empty()) return comm_fail(c, MVS_ESTATE, "mvs_comm_wait: no call in flight (mvs_comm_run_async)");
    const int slot = c->pending.front();
    c->pending.pop_front();
    if (slot >= 0)
        for (int r = 0; r < c->n; r++)
            if (c->sent_valid[r][slot]) {
                const hipError_t q = hipEventSynchronize(c->sent[r][slot]);
                if (q != hipSuccess) {
                    c->have_result = false;
                    return comm_fail(c, MVS_EHIP, "mvs_comm_wait: rank %d (device %d): %s", r, c->devices[r], hipGetErrorString(q));
                }
            }
    c->result_slot = slot;
    c->have_result = true;
    return MVS_OK;
}

int mvs_comm_pending(const mvs_comm *c) { return c ? (int)c->pending.size() : MVS_EINVAL; }

int mvs_comm_fetch(mvs_comm *c, float *depth_hw, float *cost_hw)
{
    if (!c) return MVS_EINVAL;
    if (c->broken) return comm_fail(c, MVS_ESTATE, "mvs_comm_fetch: the communicators were aborted after a failed collective; create a new mvs_comm");
    if (!c->have_result) return comm_fail(c, MVS_ESTATE, "mvs_comm_fetch: no result yet (mvs_comm_run, or mvs_comm_run_async + mvs_comm_wait), or new inputs were set since");
    // rank 0 alone: its context (or, after mvs_comm_wait, the waited call's result pair on its device) holds the maps of the whole view
    mvs_ctx *x = c->ctx[0];
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    if (hipSetDevice(c->devices[0]) != hipSuccess) return comm_fail(c, MVS_EHIP, "mvs_comm_fetch: hipSetDevice failed");
    int e = MVS_OK;
    if (c->result_slot >= 0) {
        const size_t bytes = (size_t)c->W * c->H * 4;
        if ((depth_hw && hipMemcpy(depth_hw, c->result[c->result_slot][0].ptr, bytes, hipMemcpyDeviceToHost) != hipSuccess) ||
            (cost_hw && hipMemcpy(cost_hw, c->result[c->result_slot][1].ptr, bytes, hipMemcpyDeviceToHost) != hipSuccess))
            e = comm_fail(c, MVS_EHIP, "mvs_comm_fetch: hipMemcpy of the result maps failed");
    } else if ((e = mvs_sweep_fetch(x, depth_hw, cost_hw, nullptr, nullptr))) {
        e = comm_fail(c, e, "mvs_comm_fetch: %s", mvs_last_error(x));
    }
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    return e;
}

}  // extern "C"
