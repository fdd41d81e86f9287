// loopback_rccl.hip -- TEST INFRASTRUCTURE: a stand-in for the seven RCCL entry points mvs_comm resolves with dlopen
// (csrc/comm.cpp: ncclCommInitAll / Destroy / Abort / AllReduce / ReduceScatter / AllGather / GetErrorString), so that the thread-per-rank
// code of mvs_sweep_sharded -- host barrier, plane-group pipeline on the second stream, reduce-scatter slicing, partial merge, the abort
// path -- runs with n = 2, 4, 8 ranks on a box with ONE GPU (RCCL itself refuses two ranks per device).  Selected through the existing
// MVS_RCCL_LIBRARY hook together with MVS_COMM_ALLOW_SAME_DEVICE=1; never loaded by the product otherwise, never shipped as part of it.
//
// A collective here is: wait for the caller's stream, meet the other ranks on a host barrier, sum / gather the ranks' send buffers into a
// private staging buffer with a kernel on the caller's stream, meet again (every rank has finished READING), copy the staging buffer to
// the receive buffer.  Synchronous and slow on purpose: it is a correctness vehicle (exact integer sums, the same results RCCL gives),
// not a measurement -- DESIGN.md keeps saying that no scaling curve exists.
//
// Failure injection (read at ncclCommInitAll): LOOPBACK_RCCL_FAIL="<op>:<rank>:<k>" makes the k-th call (1-based) of <op> in
// {allreduce, reducescatter, allgather} on <rank> return ncclInternalError WITHOUT meeting the others -- what a link failure looks like to
// the caller; the other ranks stay in the barrier until ncclCommAbort wakes them, and then return ncclInternalError too.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Group {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0, generation = 0;
    bool aborted = false;
    int alive = 0;  // communicators not yet destroyed / aborted
    std::vector<const void *> send;
    // failure injection
    int fail_op = -1, fail_rank = -1, fail_call = 0;

    // false: the group was aborted while (or before) waiting
    bool meet()
    {
        std::unique_lock<std::mutex> lock(m);
        if (aborted) return false;
        const int gen = generation;
        if (++waiting == n) {
            waiting = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        cv.wait(lock, [&] { return gen != generation || aborted; });
        return gen != generation;
    }
};

enum Op { OP_ALLREDUCE = 0, OP_REDUCESCATTER = 1, OP_ALLGATHER = 2 };

}  // namespace

struct ncclComm {
    Group *g = nullptr;
    int rank = 0, device = 0;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    const void **table = nullptr;  // device copy of the ranks' send pointers
    int calls[3] = {0, 0, 0};
    bool dead = false;
};

namespace {

template <typename T>
__global__ void sum_ranks(const T *const *__restrict__ send, int n, size_t offset, size_t count, T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T s = 0;
    for (int k = 0; k < n; k++) s += send[k][offset + i];
    out[i] = s;
}

__global__ void gather_ranks(const unsigned char *const *__restrict__ send, int n, size_t bytes, unsigned char *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bytes * (size_t)n) return;
    out[i] = send[i / bytes][i % bytes];
}

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    case ncclFloat16: case ncclBfloat16: return 2;
    default: return 0;
    }
}

// send: this rank's send buffer; out_bytes: what lands in recv; the staging kernel is chosen by `op`
ncclResult_t collective(Op op, const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t red, ncclComm_t c, hipStream_t stream)
{
    if (!c || !c->g || c->dead) return ncclInvalidArgument;
    Group *g = c->g;
    const int call = ++c->calls[op];
    if (g->fail_op == (int)op && g->fail_rank == c->rank && g->fail_call == call) return ncclInternalError;  // injected: never meets the others
    const size_t ts = type_size(type);
    if (!ts) return ncclInvalidArgument;
    if (op != OP_ALLGATHER && (red != ncclSum || !(type == ncclUint32 || type == ncclInt32 || type == ncclUint64 || type == ncclInt64))) return ncclInvalidArgument;
    if (hipSetDevice(c->device) != hipSuccess) return ncclUnhandledCudaError;
    const size_t out_bytes = (op == OP_ALLGATHER ? count * (size_t)g->n : count) * ts;
    if (out_bytes > c->tmp_bytes) {
        if (c->tmp) (void)hipFree(c->tmp);
        c->tmp = nullptr;
        c->tmp_bytes = 0;
        if (hipMalloc(&c->tmp, out_bytes) != hipSuccess) return ncclUnhandledCudaError;
        c->tmp_bytes = out_bytes;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;  // this rank's send data is complete
    {
        std::lock_guard<std::mutex> lock(g->m);
        g->send[c->rank] = send;
    }
    if (!g->meet()) return ncclInternalError;  // every rank's send data is complete and its pointer published
    std::vector<const void *> ptrs;
    {
        std::lock_guard<std::mutex> lock(g->m);
        ptrs = g->send;
    }
    if (hipMemcpyAsync(c->table, ptrs.data(), sizeof(void *) * g->n, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (out_bytes) {
        const unsigned blocks = (unsigned)(((op == OP_ALLGATHER ? out_bytes : count) + 255) / 256);
        const size_t offset = op == OP_REDUCESCATTER ? count * (size_t)c->rank : 0;
        if (op == OP_ALLGATHER)
            gather_ranks<<<blocks, 256, 0, stream>>>((const unsigned char *const *)c->table, g->n, count * ts, (unsigned char *)c->tmp);
        else if (ts == 4)
            sum_ranks<uint32_t><<<blocks, 256, 0, stream>>>((const uint32_t *const *)c->table, g->n, offset, count, (uint32_t *)c->tmp);
        else
            sum_ranks<unsigned long long><<<blocks, 256, 0, stream>>>((const unsigned long long *const *)c->table, g->n, offset, count, (unsigned long long *)c->tmp);
        if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!g->meet()) return ncclInternalError;  // every rank has finished reading the send buffers (they may alias receive buffers)
    if (out_bytes && hipMemcpyAsync(recv, c->tmp, out_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;  // (the copy is ordered on the caller's stream, like a real collective's completion)
}

}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1) return ncclInvalidArgument;
    Group *g = new Group();
    g->n = ndev;
    g->alive = ndev;
    g->send.assign(ndev, nullptr);
    if (const char *f = getenv("LOOPBACK_RCCL_FAIL")) {
        char op[32] = {0};
        int rank = -1, call = 0;
        if (sscanf(f, "%31[a-z]:%d:%d", op, &rank, &call) == 3) {
            g->fail_op = !strcmp(op, "allreduce") ? OP_ALLREDUCE : !strcmp(op, "reducescatter") ? OP_REDUCESCATTER : !strcmp(op, "allgather") ? OP_ALLGATHER : -1;
            g->fail_rank = rank;
            g->fail_call = call;
        }
    }
    for (int r = 0; r < ndev; r++) {
        ncclComm *c = new ncclComm();
        c->g = g;
        c->rank = r;
        c->device = devlist ? devlist[r] : r;
        if (hipSetDevice(c->device) != hipSuccess || hipMalloc((void **)&c->table, sizeof(void *) * ndev) != hipSuccess) {
            delete c;
            for (int k = 0; k < r; k++) {
                (void)hipFree(comms[k]->table);
                delete comms[k];
            }
            delete g;
            return ncclUnhandledCudaError;
        }
        // ranks on different devices read each other's buffers directly
        for (int k = 0; k < ndev; k++) {
            const int other = devlist ? devlist[k] : k;
            if (other != c->device) (void)hipDeviceEnablePeerAccess(other, 0);
        }
        comms[r] = c;
    }
    (void)hipGetLastError();
    return ncclSuccess;
}

static void release(ncclComm_t c)
{
    Group *g = c->g;
    bool last;
    {
        std::lock_guard<std::mutex> lock(g->m);
        last = --g->alive == 0;
    }
    c->dead = true;
    if (last) delete g;
    // (an aborted communicator's staging buffers are leaked: another rank's thread may still be inside a call on it)
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c || c->dead) return ncclInvalidArgument;
    (void)hipSetDevice(c->device);
    if (c->tmp) (void)hipFree(c->tmp);
    if (c->table) (void)hipFree(c->table);
    release(c);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c)
{
    if (!c || !c->g) return ncclInvalidArgument;
    Group *g = c->g;
    {
        std::lock_guard<std::mutex> lock(g->m);
        g->aborted = true;
    }
    g->cv.notify_all();
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t c, hipStream_t stream)
{
    return collective(OP_ALLREDUCE, send, recv, count, type, op, c, stream);
}

ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t type, ncclRedOp_t op, ncclComm_t c, hipStream_t stream)
{
    return collective(OP_REDUCESCATTER, send, recv, recvcount, type, op, c, stream);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t type, ncclComm_t c, hipStream_t stream)
{
    return collective(OP_ALLGATHER, send, recv, sendcount, type, ncclSum, c, stream);
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loopback: unhandled HIP error";
    case ncclInternalError: return "loopback: internal error (injected failure, or the group was aborted)";
    case ncclInvalidArgument: return "loopback: invalid argument";
    default: return "loopback: error";
    }
}

}  // extern "C"
