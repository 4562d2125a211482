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
// It also has a MULTI-PROCESS mode (ncclCommInitRank with nranks > 1: one process per rank, as bench.py runs under
// torch.distributed.run): the ranks meet in a POSIX shared-memory segment named by the unique id; a send stages its
// bytes in a file of its own (RCCL_MODEL_DIR, default /tmp) and publishes it in the (src, dst) queue, the matching
// receive copies the file to its buffer and removes it.  (Host staging on purpose: HIP IPC handles proved unreliable
// for this -- on ROCm 7.2 hipIpcGetMemHandle refuses an allocation whose address range once held an imported mapping.)
// That mode is synchronous on the host (a legal, if slow, RCCL) and turns what would be a hang -- an unmatched send
// or receive, a size mismatch -- into an error after RCCL_MODEL_TIMEOUT_S seconds (default 60).
// Never linked into or shipped with the product.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

struct World {
    int nranks = 0;
    std::vector<int> device;
};

// ---- multi-process mode: what the ranks share ---------------------------------------------------------------
constexpr int kMaxRanks = 8, kQueue = 64;
constexpr size_t kReduceWords = 1 << 16;

struct MpEntry {
    uint64_t bytes;
};
struct MpPair {                                   // FIFO of published sends src -> dst
    std::atomic<uint64_t> posted, consumed;
    MpEntry e[kQueue];
};
struct MpShared {
    std::atomic<int> ready, bar_count, bar_gen, failed;
    MpPair pair[kMaxRanks][kMaxRanks];
    uint32_t red[kMaxRanks][kReduceWords];
};
struct Mp {
    MpShared *sh = nullptr;
    int rank = 0, nranks = 0;
    double timeout_s = 60.0;
    std::string stage;                            // prefix of the staging files: <dir>/<id>
};

static std::string stage_name(const Mp *mp, int src, int dst, uint64_t seq)
{
    char tail[64];
    std::snprintf(tail, sizeof tail, "-%d-%d-%llu", src, dst, (unsigned long long)seq);
    return mp->stage + tail;
}

// device <-> staging file; `to_file` creates it
static bool stage_copy(const std::string &name, void *dev, size_t bytes, bool to_file)
{
    const int fd = open(name.c_str(), to_file ? (O_CREAT | O_TRUNC | O_RDWR) : O_RDONLY, 0600);
    if (fd < 0) return false;
    bool ok = !to_file || ftruncate(fd, (off_t)bytes) == 0;
    void *m = ok && bytes ? mmap(nullptr, bytes, to_file ? (PROT_READ | PROT_WRITE) : PROT_READ, MAP_SHARED, fd, 0) : nullptr;
    close(fd);
    if (bytes) {
        ok = ok && m != MAP_FAILED;
        if (ok) ok = hipMemcpy(to_file ? m : dev, to_file ? dev : m, bytes, to_file ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice) == hipSuccess;
        if (m && m != MAP_FAILED) munmap(m, bytes);
    }
    if (!to_file) unlink(name.c_str());
    return ok;
}

struct Op {
    int kind;            // 0 send, 1 recv, 2 allreduce
    int rank, peer;
    void *buf, *buf2;
    size_t bytes;
    hipStream_t stream;
    World *world;
    Mp *mp;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
int g_errors = 0;

}  // namespace

struct ncclComm {
    World *world;
    int rank;
    Mp *mp;
};

// ---- multi-process group ------------------------------------------------------------------------------------
template <class F>
static bool mp_wait(Mp *mp, const char *what, F done)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; !done(); ++spin) {
        if (mp->sh->failed.load()) return false;
        if (spin > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > mp->timeout_s) {
            std::fprintf(stderr, "rccl_model[rank %d]: %s: nothing for %.0f s -- real RCCL would hang here\n", mp->rank, what, mp->timeout_s);
            mp->sh->failed.store(1);
            ++g_errors;
            return false;
        }
    }
    return true;
}

static bool mp_barrier(Mp *mp)
{
    MpShared *sh = mp->sh;
    const int gen = sh->bar_gen.load();
    if (sh->bar_count.fetch_add(1) + 1 == mp->nranks) {
        sh->bar_count.store(0);
        sh->bar_gen.fetch_add(1);
        return true;
    }
    return mp_wait(mp, "barrier", [&] { return sh->bar_gen.load() != gen; });
}

static ncclResult_t run_group_mp(std::vector<Op> &ops)
{
    Mp *mp = ops[0].mp;
    MpShared *sh = mp->sh;
    for (const Op &o : ops) {
        if (o.mp != mp) {
            std::fprintf(stderr, "rccl_model: one communicator per group in multi-process mode\n");
            return ncclInvalidUsage;
        }
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;   // sources produced, destinations free
    }
    // 1. publish every send (FIFO per destination); nothing here waits on a peer
    std::vector<std::pair<int, uint64_t>> mine;                           // (dst, sequence number) to see consumed
    for (const Op &o : ops) {
        if (o.kind != 0) continue;
        MpPair &q = sh->pair[mp->rank][o.peer];
        const uint64_t seq = q.posted.load();
        if (seq - q.consumed.load() >= (uint64_t)kQueue) {
            std::fprintf(stderr, "rccl_model[rank %d]: more than %d sends in flight to rank %d\n", mp->rank, kQueue, o.peer);
            return ncclInternalError;
        }
        MpEntry &e = q.e[seq % kQueue];
        if (!stage_copy(stage_name(mp, mp->rank, o.peer, seq), o.buf, o.bytes, true)) {
            std::fprintf(stderr, "rccl_model[rank %d]: staging a send of %zu bytes failed\n", mp->rank, o.bytes);
            sh->failed.store(1);
            return ncclSystemError;
        }
        e.bytes = o.bytes;
        q.posted.store(seq + 1, std::memory_order_release);
        mine.emplace_back(o.peer, seq);
    }
    // 2. every receive, in order per source
    for (const Op &o : ops) {
        if (o.kind != 1) continue;
        MpPair &q = sh->pair[o.peer][mp->rank];
        const uint64_t seq = q.consumed.load();
        char what[96];
        std::snprintf(what, sizeof what, "receive of %zu bytes from rank %d without a send", o.bytes, o.peer);
        if (!mp_wait(mp, what, [&] { return q.posted.load(std::memory_order_acquire) > seq; })) return ncclInternalError;
        const MpEntry &e = q.e[seq % kQueue];
        if (e.bytes != o.bytes) {
            std::fprintf(stderr, "rccl_model[rank %d]: send of %llu bytes from rank %d meets a receive of %zu\n", mp->rank, (unsigned long long)e.bytes,
                         o.peer, o.bytes);
            sh->failed.store(1);
            ++g_errors;
            return ncclInvalidArgument;
        }
        if (!stage_copy(stage_name(mp, o.peer, mp->rank, seq), o.buf, o.bytes, false)) {
            std::fprintf(stderr, "rccl_model[rank %d]: reading the staged send from rank %d failed\n", mp->rank, o.peer);
            sh->failed.store(1);
            return ncclSystemError;
        }
        q.consumed.store(seq + 1, std::memory_order_release);
    }
    // 3. a send is complete when its data has left
    for (auto &m : mine) {
        MpPair &q = sh->pair[mp->rank][m.first];
        char what[96];
        std::snprintf(what, sizeof what, "send to rank %d without a receive", m.first);
        if (!mp_wait(mp, what, [&] { return q.consumed.load(std::memory_order_acquire) > m.second; })) return ncclInternalError;
    }
    // all-reduce: uint32 sum through the shared segment
    for (const Op &o : ops) {
        if (o.kind != 2) continue;
        const size_t n = o.bytes / 4;
        if (n > kReduceWords) return ncclInvalidArgument;
        if (hipMemcpy(sh->red[mp->rank], o.buf, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!mp_barrier(mp)) return ncclInternalError;
        std::vector<uint32_t> sum(n, 0);
        for (int r = 0; r < mp->nranks; ++r)
            for (size_t k = 0; k < n; ++k) sum[k] += sh->red[r][k];
        if (hipMemcpy(o.buf2, sum.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        if (!mp_barrier(mp)) return ncclInternalError;                   // red[] may be overwritten after this
    }
    return ncclSuccess;
}

static ncclResult_t run_group()
{
    std::vector<Op> ops;
    ops.swap(g_ops);
    if (!ops.empty() && ops[0].mp) return run_group_mp(ops);
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
    static std::atomic<unsigned> serial{0};
    std::snprintf(reinterpret_cast<char *>(id), sizeof *id, "/x266-rccl-model-%d-%u-%llx", (int)getpid(), serial.fetch_add(1),
                  (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    World *w = new World;
    w->nranks = ndev;
    for (int i = 0; i < ndev; ++i) w->device.push_back(devlist ? devlist[i] : i);
    for (int i = 0; i < ndev; ++i) comms[i] = new ncclComm{w, i, nullptr};
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    World *w = new World;
    w->nranks = nranks;
    int dev = 0;
    hipGetDevice(&dev);
    w->device.assign((size_t)nranks, dev);
    if (nranks == 1 && rank == 0) {
        *comm = new ncclComm{w, 0, nullptr};
        return ncclSuccess;
    }
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    // one process per rank: meet in the shared segment the id names
    char name[sizeof id + 1];
    std::memcpy(name, &id, sizeof id);
    name[sizeof id] = 0;
    if (name[0] != '/') return ncclInvalidArgument;
    Mp *mp = new Mp;
    mp->rank = rank;
    mp->nranks = nranks;
    if (const char *t = std::getenv("RCCL_MODEL_TIMEOUT_S")) mp->timeout_s = std::atof(t);
    const char *dir = std::getenv("RCCL_MODEL_DIR");
    mp->stage = std::string(dir && *dir ? dir : "/tmp") + name;
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    const bool creator = fd >= 0;
    if (!creator) {
        const auto t0 = std::chrono::steady_clock::now();
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > mp->timeout_s) return ncclSystemError;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    if (creator && ftruncate(fd, sizeof(MpShared)) != 0) return ncclSystemError;
    if (!creator) {                                                      // the creator sizes it before anyone maps it
        struct stat st;
        const auto t0 = std::chrono::steady_clock::now();
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < sizeof(MpShared)) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > mp->timeout_s) return ncclSystemError;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    void *m = mmap(nullptr, sizeof(MpShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    mp->sh = static_cast<MpShared *>(m);                                 // a fresh segment is all zeroes: every atomic starts at 0
    if (creator) mp->sh->ready.store(1, std::memory_order_release);
    if (!mp_wait(mp, "communicator set-up", [&] { return mp->sh->ready.load(std::memory_order_acquire) == 1; }) || !mp_barrier(mp)) return ncclInternalError;
    if (creator) shm_unlink(name);                                       // everyone has it mapped
    *comm = new ncclComm{w, rank, mp};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (comm && comm->mp) {
        Mp *mp = comm->mp;
        for (int dst = 0; dst < mp->nranks; ++dst) {                     // staging files nobody came for (a failed run)
            MpPair &q = mp->sh->pair[mp->rank][dst];
            for (uint64_t seq = q.consumed.load(); seq < q.posted.load(); ++seq) unlink(stage_name(mp, mp->rank, dst, seq).c_str());
        }
        munmap(comm->mp->sh, sizeof(MpShared));
        delete comm->mp;
    }
    delete comm;                                                         // the World is shared and small: left to the process
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }      // nothing is ever in flight here: every transfer is host-synchronous

ncclResult_t ncclGetVersion(int *version)
{
    if (version) *version = 0;                                           // "the model", not a release of RCCL
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
    g_ops.push_back(Op{kind, comm->rank, peer, buf, buf2, bytes, stream, comm->world, comm->mp});
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
