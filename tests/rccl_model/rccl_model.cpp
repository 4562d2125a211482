// rccl_model.cpp -- TEST INFRASTRUCTURE: a single-process model of the RCCL calls libx266hip's node layer makes,
// so that the layer's RCCL transport (the ncclGroupStart/End of ncclSend/ncclRecv it builds per step) can run with
// N > 1 ranks on the ONE GPU of a test box.  Real RCCL refuses two ranks on one device; the node layer then falls back
// to peer copies, which leaves exactly its RCCL code path untested there.  This model is loaded INSTEAD of librccl
// (X266HIP_RCCL_LIB=<path>) and enforces the semantics that matter for correctness:
//   * inside a group, the k-th ncclSend from rank a to rank b pairs with the k-th ncclRecv on b from a, and their
//     byte counts must be equal -- anything unmatched at the outermost ncclGroupEnd is an error here (it would be a
//     hang with real RCCL), so an ordering or sizing bug between the root's and a peer's lists cannot pass;
//   * a transfer is ordered after everything already enqueued on BOTH ranks' streams, and later work on both streams
//     is ordered after it (events), like a send/recv kernel pair;
//   * ncclAllReduce (uint32 sum only) over all ranks of the communicator.
// Never linked into or shipped with the product.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

namespace {

struct World {
    int nranks = 0;
    std::vector<int> device;
};

struct Op {
    int kind;            // 0 send, 1 recv, 2 allreduce
    int rank, peer;
    void *buf, *buf2;
    size_t bytes;
    hipStream_t stream;
    World *world;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
int g_errors = 0;

}  // namespace

struct ncclComm {
    World *world;
    int rank;
};

static ncclResult_t run_group()
{
    std::vector<Op> ops;
    ops.swap(g_ops);
    // pair sends and receives per (src, dst), FIFO
    std::map<std::pair<int, int>, std::vector<size_t>> sends, recvs;
    std::vector<size_t> reduces;
    for (size_t i = 0; i < ops.size(); ++i) {
        if (ops[i].kind == 0) sends[{ops[i].rank, ops[i].peer}].push_back(i);
        else if (ops[i].kind == 1) recvs[{ops[i].peer, ops[i].rank}].push_back(i);
        else reduces.push_back(i);
    }
    for (auto &kv : sends) {
        auto &rv = recvs[kv.first];
        if (rv.size() != kv.second.size()) {
            std::fprintf(stderr, "rccl_model: %zu send(s) %d -> %d but %zu receive(s): real RCCL would hang\n", kv.second.size(), kv.first.first,
                         kv.first.second, rv.size());
            ++g_errors;
            return ncclInternalError;
        }
        for (size_t k = 0; k < rv.size(); ++k) {
            const Op &s = ops[kv.second[k]], &r = ops[rv[k]];
            if (s.bytes != r.bytes) {
                std::fprintf(stderr, "rccl_model: transfer %zu of %d -> %d: send of %zu bytes meets receive of %zu\n", k, s.rank, r.rank, s.bytes, r.bytes);
                ++g_errors;
                return ncclInvalidArgument;
            }
            const int sdev = s.world->device[(size_t)s.rank], rdev = r.world->device[(size_t)r.rank];
            hipEvent_t ready, done;
            hipSetDevice(sdev);
            hipEventCreateWithFlags(&ready, hipEventDisableTiming);
            hipEventRecord(ready, s.stream);                              // the source data is produced on the sender's stream
            hipSetDevice(rdev);
            hipStreamWaitEvent(r.stream, ready, 0);
            if (hipMemcpyPeerAsync(r.buf, rdev, s.buf, sdev, s.bytes, r.stream) != hipSuccess) return ncclUnhandledCudaError;
            hipEventCreateWithFlags(&done, hipEventDisableTiming);
            hipEventRecord(done, r.stream);
            hipSetDevice(sdev);
            hipStreamWaitEvent(s.stream, done, 0);                        // the sender may reuse its buffer only after the data has left
            hipEventDestroy(ready);
            hipEventDestroy(done);
        }
        recvs.erase(kv.first);
    }
    for (auto &kv : recvs)
        if (!kv.second.empty()) {
            std::fprintf(stderr, "rccl_model: %zu receive(s) on %d from %d without a send: real RCCL would hang\n", kv.second.size(), kv.first.second, kv.first.first);
            ++g_errors;
            return ncclInternalError;
        }
    if (!reduces.empty()) {                                              // uint32 sum over all ranks of the world; host arithmetic (test sizes)
        World *w = ops[reduces[0]].world;
        if ((int)reduces.size() != w->nranks) {
            std::fprintf(stderr, "rccl_model: all-reduce posted by %zu of %d ranks\n", reduces.size(), w->nranks);
            ++g_errors;
            return ncclInternalError;
        }
        const size_t n = ops[reduces[0]].bytes / 4;
        std::vector<uint32_t> sum(n, 0), tmp(n);
        for (size_t i : reduces) {
            hipSetDevice(ops[i].world->device[(size_t)ops[i].rank]);
            hipStreamSynchronize(ops[i].stream);
            hipMemcpy(tmp.data(), ops[i].buf, n * 4, hipMemcpyDeviceToHost);
            for (size_t k = 0; k < n; ++k) sum[k] += tmp[k];
        }
        for (size_t i : reduces) {
            hipSetDevice(ops[i].world->device[(size_t)ops[i].rank]);
            hipMemcpy(ops[i].buf2, sum.data(), n * 4, hipMemcpyHostToDevice);
        }
    }
    return ncclSuccess;
}

extern "C" {

int rccl_model_errors(void) { return g_errors; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof *id);
    std::memcpy(id, "x266-rccl-model", 16);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    World *w = new World;
    w->nranks = ndev;
    for (int i = 0; i < ndev; ++i) w->device.push_back(devlist ? devlist[i] : i);
    for (int i = 0; i < ndev; ++i) comms[i] = new ncclComm{w, i};
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank)
{
    if (nranks != 1 || rank != 0) {
        std::fprintf(stderr, "rccl_model: one process only (ncclCommInitRank with %d ranks)\n", nranks);
        return ncclInvalidUsage;
    }
    World *w = new World;
    w->nranks = 1;
    int dev = 0;
    hipGetDevice(&dev);
    w->device.push_back(dev);
    *comm = new ncclComm{w, 0};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete comm;                                                         // the World is shared and small: left to the process
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void)
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    int prev = 0;
    hipGetDevice(&prev);
    const ncclResult_t r = run_group();
    hipSetDevice(prev);
    return r;
}

static ncclResult_t post(int kind, void *buf, void *buf2, size_t bytes, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || (kind != 2 && (peer < 0 || peer >= comm->world->nranks))) return ncclInvalidArgument;
    g_ops.push_back(Op{kind, comm->rank, peer, buf, buf2, bytes, stream, comm->world});
    if (g_depth == 0) {                                                  // outside a group: an implicit group of one
        ++g_depth;
        return ncclGroupEnd();
    }
    return ncclSuccess;
}

static size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 2;
    }
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(0, const_cast<void *>(buf), nullptr, count * type_bytes(t), peer, comm, stream);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(1, buf, nullptr, count * type_bytes(t), peer, comm, stream);
}

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    if (t != ncclUint32 || op != ncclSum) return ncclInvalidArgument;
    return post(2, const_cast<void *>(sendbuf), recvbuf, count * 4, -1, comm, stream);
}

const char *ncclGetErrorString(ncclResult_t r)
{
    return r == ncclSuccess ? "no error" : (r == ncclInternalError ? "rccl_model: unmatched send/recv (would hang)" : "rccl_model: invalid use");
}

}  // extern "C"
