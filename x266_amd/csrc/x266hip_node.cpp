// x266hip_node.cpp -- one node, several GPUs: the scatter -> transform -> gather path of
// BASELINE configs[4] behind the C ABI (include/x266hip.h, "one node, several GPUs").
//
// The reference has no multi-device code at all; what makes the path shard is
// that every block is independent (src_tb/dct32.c:75,167-168: partialButterfly32 keeps no state
// between rows beyond the block; satd8x8, src_tb/satd.c:31-118, is a pure function).  So the only
// traffic is moving shards between the root's HBM and the peers', and it is issued the way xGMI wants
// it: per step ONE ncclGroupStart/End holding every ncclSend/ncclRecv of that step -- root -> peers
// inputs of frame t and peers -> root results of frame t-2 -- so all seven links of the root and both
// directions of each carry data at once, on a communication stream of its own, ordered against the
// kernels by events only.  No collective is needed (or used) on the data path.
//
// Built purely on the public device-pointer ABI of this library (xDct32FwdBatchDev, ...), the HIP
// runtime and RCCL; RCCL is opened with dlopen on first use so that the BDPI drop-in does not pull it in.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/x266hip.h"

namespace {

// ---- RCCL, resolved at run time --------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;       // optional
    decltype(&ncclGetVersion) GetVersion = nullptr;     // optional
    std::string why;
};

void load_rccl(Rccl &r)
{
    // X266HIP_RCCL_LIB (documented in include/x266hip.h) names the library to use instead: another RCCL build, or the
    // tests' single-box RCCL model.  Which library was loaded is reported by xHipNodeRcclInfo.
    const char *names[] = {std::getenv("X266HIP_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        const char *why = dlerror();
        r.why = std::string("dlopen(librccl.so.1): ") + (why ? why : "not found");
        return;
    }
    bool ok = true;
#define X_SYM(field, name) ok = ok && ((r.field = (decltype(r.field))dlsym(r.handle, name)) != nullptr)
    X_SYM(GetUniqueId, "ncclGetUniqueId");
    X_SYM(CommInitRank, "ncclCommInitRank");
    X_SYM(CommInitAll, "ncclCommInitAll");
    X_SYM(CommDestroy, "ncclCommDestroy");
    X_SYM(GroupStart, "ncclGroupStart");
    X_SYM(GroupEnd, "ncclGroupEnd");
    X_SYM(Send, "ncclSend");
    X_SYM(Recv, "ncclRecv");
    X_SYM(AllReduce, "ncclAllReduce");
    X_SYM(GetErrorString, "ncclGetErrorString");
#undef X_SYM
    if (ok) {
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.handle, "ncclCommAbort");
        r.GetVersion = (decltype(r.GetVersion))dlsym(r.handle, "ncclGetVersion");
    }
    if (!ok) {
        r.why = "librccl.so.1 lacks an expected symbol";
        dlclose(r.handle);
        r.handle = nullptr;
    }
}

Rccl &rccl_state()                         // loaded once, whichever thread asks first
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, load_rccl, std::ref(r));
    return r;
}

Rccl *rccl()
{
    Rccl &r = rccl_state();
    return r.handle ? &r : nullptr;
}

struct DeviceScope {                       // the caller's current device is put back on exit
    int prev = -1;
    explicit DeviceScope(int dev) { (void)hipGetDevice(&prev); (void)hipSetDevice(dev); }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct DevBuf {                            // grow-only device allocation
    void *p = nullptr;
    size_t bytes = 0;
};

struct LocalRank {
    int device = 0;
    int rank = 0;
    x266hip_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    static constexpr int kComputeStreams = 3;  // = x266hip_nstream::kSlots: frame t runs on stream t % 3 -- consecutive frames share nothing, so their kernels may overlap
    hipStream_t comm_stream = nullptr;
    hipStream_t compute[kComputeStreams] = {};  // [0] also carries everything that is not a frame stream (motion search stripes)
    std::vector<DevBuf> me_bufs;           // motion-search stripe buffers (cur, ref, best per owned stripe)
    DevBuf selftest;
};

const size_t kInUnit[3] = {2048, 2048, 128};
const size_t kOutUnit[3] = {2048, 2048, 4};

}  // namespace

struct x266hip_node {
    int world = 1;
    int root = 0;
    bool single_process = true;
    bool have_rccl = false;
    int transport = 0;                     // 0 RCCL groups, 1 peer copies (single process only)
    int me_local_copy = 0;
    int fused_frame_lanes = 1;             // frame streams (DCT32 forward + SATD lanes): one launch per frame and rank
    bool failed = false;                   // a communication step failed: the communicators were aborted, the node only remains to be freed
    struct x266hip_nstream *sg_stream[3] = {};   // xNodeBatchScatterGather's internal stream per op (grow-only, freed with the node)
    size_t sg_cap[3] = {};
    std::vector<LocalRank> local;
    std::string err;
};

struct x266hip_nstream {
    x266hip_node *node = nullptr;
    int n_lanes = 0;
    int op[4] = {};
    size_t max_units[4] = {};
    // three slots: a frame's kernels take ~30 us, and only a third frame in flight covers the ramp and tail of the two before it
    // (one-rank 8K stream: 36.8 -> 34.3 us per frame); kRing >= kSlots + 3 step records
    static constexpr int kSlots = 3, kRing = 6;
    static_assert(X266_STREAM_IN_RING == kSlots + 1 && X266_STREAM_OUT_RING == kSlots + 2 && kRing >= X266_STREAM_OUT_RING, "ring sizes of the header follow the slots");
    static_assert(kSlots == LocalRank::kComputeStreams, "one compute stream per slot");
    struct Step {
        bool has_frame = false;
        size_t units[4] = {};
        const char *d_in[4] = {};
        char *d_out[4] = {};
    } ring[kRing];
    struct PerRank {
        void *in[kSlots][4] = {};          // peers only: shard slot buffers
        void *out[kSlots][4] = {};
        hipEvent_t ev_xfer[kSlots] = {}, ev_done[kSlots] = {};
        bool xfer_recorded[kSlots] = {}, done_recorded[kSlots] = {};
        hipEvent_t ev_producer = nullptr;
    };
    std::vector<PerRank> per;
    long n_steps = 0;
    long flushed_steps = 0;                // steps [0, flushed_steps) are complete: their buffers belong to the caller again
};

namespace {

int nfail(x266hip_node *n, int code, const std::string &what)
{
    if (n) n->err = what;
    return code;
}

#define N_HIP(node, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return nfail((node), X266HIP_EDEVICE, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)

#define N_NCCL(node, call)                                                                        \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) return nfail((node), X266HIP_ECOMM, std::string(#call ": ") + rccl()->GetErrorString(r_)); \
    } while (0)

#define N_X(node, lr, call)                                                                       \
    do {                                                                                          \
        int rc_ = (call);                                                                         \
        if (rc_ != X266HIP_OK) return nfail((node), rc_, std::string(#call ": ") + xHipLastError((lr).ctx)); \
    } while (0)

void shard(size_t n, int rank, int world, size_t *b, size_t *e)
{
    const size_t base = n / (size_t)world, extra = n % (size_t)world;
    const size_t r = (size_t)rank;
    *b = r * base + (r < extra ? r : extra);
    *e = *b + base + (r < extra ? 1 : 0);
}

int grow(x266hip_node *node, DevBuf &b, size_t bytes)
{
    if (bytes <= b.bytes) return X266HIP_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    if (hipMalloc(&b.p, bytes) != hipSuccess) return nfail(node, X266HIP_ENOMEM, "hipMalloc (node buffer)");
    b.bytes = bytes;
    return X266HIP_OK;
}

int open_rank(x266hip_node *node, LocalRank &lr)
{
    int rc = xHipCodecInit(&lr.ctx, lr.device);
    if (rc != X266HIP_OK) return nfail(node, rc, "xHipCodecInit failed for a node device");
    DeviceScope dev(lr.device);
    N_HIP(node, hipStreamCreateWithFlags(&lr.comm_stream, hipStreamNonBlocking));
    for (hipStream_t &cs : lr.compute) N_HIP(node, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    return X266HIP_OK;
}

// A step failed on a multi-rank node: peers may sit in an RCCL group that will never complete.  Abort the communicators
// (ncclCommAbort, when the library has it) so that neither they nor this process's stream synchronisations hang; the
// node is then good for xHipNodeFree only.  Every rank must treat a failed step the same way.
void abort_comms(x266hip_node *node)
{
    if (!node || node->failed || node->world < 2) return;
    node->failed = true;
    Rccl *R = rccl();
    if (!R || !R->CommAbort) return;
    for (LocalRank &lr : node->local)
        if (lr.comm) { (void)R->CommAbort(lr.comm); lr.comm = nullptr; }
}

LocalRank *root_rank(x266hip_node *node)
{
    for (LocalRank &lr : node->local)
        if (lr.rank == node->root) return &lr;
    return nullptr;
}

int launch(x266hip_node *node, LocalRank &lr, int op, const void *in, void *out, size_t n, hipStream_t stream)
{
    if (n == 0) return X266HIP_OK;
    switch (op) {
    case 0: N_X(node, lr, xDct32FwdBatchDev(lr.ctx, (const int16_t *)in, (int16_t *)out, n, stream)); break;
    case 1: N_X(node, lr, xDct32InvBatchDev(lr.ctx, (const int16_t *)in, (int16_t *)out, n, stream)); break;
    default: N_X(node, lr, xSatd8x8BatchDev(lr.ctx, (const int16_t *)in, (uint32_t *)out, n, stream)); break;
    }
    return X266HIP_OK;
}

// One transfer of a step as seen from one rank.  Both transports consume the same list.
struct Xfer {
    int local;            // index into node->local of the rank that posts it
    bool send;            // send to / receive from `peer`
    int peer;             // global rank
    void *buf;
    size_t bytes;
};

// Issues the step's transfers.  RCCL: one group with everything.  Peer copies (single process): each
// (root, peer) pair becomes one hipMemcpyPeerAsync on the PEER's communication stream, built from the
// peer-side entry and its matching root-side entry (same order on both sides by construction).
int post_transfers(x266hip_node *node, const std::vector<Xfer> &xs)
{
    if (node->failed) return nfail(node, X266HIP_ECOMM, "an earlier step of this node failed and its communicators were aborted");
    if (xs.empty()) return X266HIP_OK;
    if (node->transport == 0) {
        if (!node->have_rccl) return nfail(node, X266HIP_ECOMM, "RCCL did not initialise on this node");
        Rccl *R = rccl();
        (void)hipGetLastError();                                          // the host's stale error is not this group's (see xHipNodeInit)
        N_NCCL(node, R->GroupStart());
        for (const Xfer &x : xs) {
            LocalRank &lr = node->local[(size_t)x.local];
            ncclResult_t r = x.send ? R->Send(x.buf, x.bytes, ncclUint8, x.peer, lr.comm, lr.comm_stream)
                                    : R->Recv(x.buf, x.bytes, ncclUint8, x.peer, lr.comm, lr.comm_stream);
            if (r != ncclSuccess) {
                (void)R->GroupEnd();
                return nfail(node, X266HIP_ECOMM, std::string("ncclSend/ncclRecv: ") + R->GetErrorString(r));
            }
        }
        N_NCCL(node, R->GroupEnd());
        return X266HIP_OK;
    }
    if (!node->single_process) return nfail(node, X266HIP_EINVAL, "transport 1 (peer copies) needs a single-process node");
    // pair every peer-side entry with the root-side entry of the same (peer, direction), in order
    std::vector<bool> used(xs.size(), false);
    for (size_t i = 0; i < xs.size(); ++i) {
        const Xfer &p = xs[i];
        LocalRank &plr = node->local[(size_t)p.local];
        if (plr.rank == node->root) continue;
        size_t j = 0;
        for (; j < xs.size(); ++j) {
            const Xfer &q = xs[j];
            if (used[j] || node->local[(size_t)q.local].rank != node->root) continue;
            if (q.peer == plr.rank && q.send != p.send && q.bytes == p.bytes) break;
        }
        if (j == xs.size()) return nfail(node, X266HIP_ECOMM, "unmatched transfer in a step (internal)");
        used[j] = true;
        const Xfer &q = xs[j];
        LocalRank &rlr = node->local[(size_t)q.local];
        DeviceScope dev(plr.device);
        if (p.send) N_HIP(node, hipMemcpyPeerAsync(q.buf, rlr.device, p.buf, plr.device, p.bytes, plr.comm_stream));
        else        N_HIP(node, hipMemcpyPeerAsync(p.buf, plr.device, q.buf, rlr.device, p.bytes, plr.comm_stream));
    }
    return X266HIP_OK;
}

}  // namespace

extern "C" {

// ---- host-only planning ------------------------------------------------------------------------
int xShardRange(size_t n_units, int rank, int world, size_t *begin, size_t *end)
{
    if (world < 1 || rank < 0 || rank >= world || !begin || !end) return X266HIP_EINVAL;
    shard(n_units, rank, world, begin, end);
    return X266HIP_OK;
}

int xMeStripePlan(int height, int range, int stripe, int n_stripes, int *block_row_begin, int *block_row_end,
                  int *ref_row_begin, int *ref_row_end)
{
    if (height < 8 || (height & 7) || range < 0 || n_stripes < 1 || stripe < 0 || stripe >= n_stripes) return X266HIP_EINVAL;
    size_t b, e;
    shard((size_t)(height / 8), stripe, n_stripes, &b, &e);
    if (block_row_begin) *block_row_begin = (int)b;
    if (block_row_end) *block_row_end = (int)e;
    if (ref_row_begin) *ref_row_begin = (int)b * 8 - range;        // rows of the padded reference, frame coordinates
    if (ref_row_end) *ref_row_end = (int)e * 8 + range;
    return X266HIP_OK;
}

// ---- node ---------------------------------------------------------------------------------------
void xHipNodeFree(x266hip_node *node)
{
    if (!node) return;
    for (x266hip_nstream *&sg : node->sg_stream)
        if (sg) { xNodeStreamFree(sg); sg = nullptr; }
    Rccl *R = rccl();
    for (LocalRank &lr : node->local) {
        DeviceScope dev(lr.device);
        if (lr.comm_stream) (void)hipStreamSynchronize(lr.comm_stream);
        for (hipStream_t cs : lr.compute) if (cs) (void)hipStreamSynchronize(cs);
        if (lr.comm && R) (void)R->CommDestroy(lr.comm);
        for (DevBuf &b : lr.me_bufs)
            if (b.p) (void)hipFree(b.p);
        if (lr.selftest.p) (void)hipFree(lr.selftest.p);
        if (lr.comm_stream) (void)hipStreamDestroy(lr.comm_stream);
        for (hipStream_t cs : lr.compute) if (cs) (void)hipStreamDestroy(cs);
        if (lr.ctx) xHipCodecFree(lr.ctx);
    }
    delete node;
}

int xHipNodeInit(x266hip_node **out, const int *devices, int n_devices)
{
    if (!out || n_devices < 1) return X266HIP_EINVAL;
    *out = nullptr;
    x266hip_node *node = new (std::nothrow) x266hip_node;
    if (!node) return X266HIP_ENOMEM;
    node->world = n_devices;
    node->single_process = true;
    node->local.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) {
        node->local[(size_t)i].device = devices ? devices[i] : i;
        node->local[(size_t)i].rank = i;
        int rc = open_rank(node, node->local[(size_t)i]);
        if (rc != X266HIP_OK) {
            std::fprintf(stderr, "x266hip: node device %d: %s\n", node->local[(size_t)i].device, node->err.c_str());
            xHipNodeFree(node);
            return rc;
        }
    }
    Rccl *R = rccl();
    if (R) {
        std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
        std::vector<int> devs((size_t)n_devices);
        for (int i = 0; i < n_devices; ++i) devs[(size_t)i] = node->local[(size_t)i].device;
        int prev = -1;
        (void)hipGetDevice(&prev);
        // the runtime's sticky "last error" belongs to whatever call of the host failed last (a refused launch, a failed allocation):
        // cleared here so that a library which polls hipGetLastError() cannot mistake it for its own failure
        (void)hipGetLastError();
        const ncclResult_t r = R->CommInitAll(comms.data(), n_devices, devs.data());
        if (prev >= 0) (void)hipSetDevice(prev);
        if (r == ncclSuccess) {
            for (int i = 0; i < n_devices; ++i) node->local[(size_t)i].comm = comms[(size_t)i];
            node->have_rccl = true;
        } else {
            node->err = std::string("ncclCommInitAll: ") + R->GetErrorString(r);
        }
    } else {
        node->err = "RCCL unavailable: " + rccl_state().why;
    }
    if (!node->have_rccl) {                 // a single-process node can still move shards with peer copies
        std::fprintf(stderr, "x266hip: node falls back to hipMemcpyPeerAsync transport (%s)\n", node->err.c_str());
        node->transport = 1;
    }
    *out = node;
    return X266HIP_OK;
}

int xHipNodeUniqueId(void *id)
{
    if (!id) return X266HIP_EINVAL;
    Rccl *R = rccl();
    if (!R) return X266HIP_ECOMM;
    static_assert(sizeof(ncclUniqueId) == X266HIP_NODE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return X266HIP_ECOMM;
    std::memcpy(id, &u, sizeof u);
    return X266HIP_OK;
}

int xHipNodeInitRank(x266hip_node **out, int device, int rank, int world, const void *id)
{
    if (!out || world < 1 || rank < 0 || rank >= world || !id) return X266HIP_EINVAL;
    *out = nullptr;
    Rccl *R = rccl();
    if (!R) {
        std::fprintf(stderr, "x266hip: RCCL is required for a process-per-GPU node and could not be loaded (%s)\n", rccl_state().why.c_str());
        return X266HIP_ECOMM;
    }
    x266hip_node *node = new (std::nothrow) x266hip_node;
    if (!node) return X266HIP_ENOMEM;
    node->world = world;
    node->single_process = false;
    node->local.resize(1);
    node->local[0].device = device;
    node->local[0].rank = rank;
    int rc = open_rank(node, node->local[0]);
    if (rc != X266HIP_OK) {
        std::fprintf(stderr, "x266hip: node rank %d, device %d: %s\n", rank, device, node->err.c_str());
        xHipNodeFree(node);
        return rc;
    }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    {
        DeviceScope dev(device);
        (void)hipGetLastError();                                          // see xHipNodeInit: the host's stale error is not this call's
        const ncclResult_t r = R->CommInitRank(&node->local[0].comm, world, u, rank);
        if (r != ncclSuccess) {
            std::fprintf(stderr, "x266hip: ncclCommInitRank(rank %d of %d): %s\n", rank, world, R->GetErrorString(r));
            xHipNodeFree(node);
            return X266HIP_ECOMM;
        }
    }
    node->have_rccl = true;
    *out = node;
    return X266HIP_OK;
}

int xHipNodeWorld(const x266hip_node *node) { return node ? node->world : 0; }
int xHipNodeLocalCount(const x266hip_node *node) { return node ? (int)node->local.size() : 0; }
int xHipNodeLocalRank(const x266hip_node *node, int i)
{
    return (node && i >= 0 && (size_t)i < node->local.size()) ? node->local[(size_t)i].rank : -1;
}
x266hip_ctx *xHipNodeCtx(x266hip_node *node, int i)
{
    return (node && i >= 0 && (size_t)i < node->local.size()) ? node->local[(size_t)i].ctx : nullptr;
}
const char *xHipNodeLastError(const x266hip_node *node) { return node ? node->err.c_str() : "null node"; }

int xHipNodeSetOption(x266hip_node *node, const char *key, int value)
{
    if (!node || !key) return X266HIP_EINVAL;
    if (!std::strcmp(key, "transport")) {
        if (value == 0 && !node->have_rccl) return nfail(node, X266HIP_ECOMM, "transport 0 needs RCCL, which did not initialise");
        if (value == 1 && !node->single_process) return nfail(node, X266HIP_EINVAL, "transport 1 (peer copies) needs a single-process node");
        if (value != 0 && value != 1) return nfail(node, X266HIP_EINVAL, "transport must be 0 or 1");
        node->transport = value;
        return X266HIP_OK;
    }
    if (!std::strcmp(key, "fused_frame_lanes")) {
        if (value != 0 && value != 1) return nfail(node, X266HIP_EINVAL, "fused_frame_lanes must be 0 or 1");
        node->fused_frame_lanes = value;
        return X266HIP_OK;
    }
    if (!std::strcmp(key, "me_local_copy")) {
        if (value != 0 && value != 1) return nfail(node, X266HIP_EINVAL, "me_local_copy must be 0 or 1");
        node->me_local_copy = value;
        return X266HIP_OK;
    }
    return nfail(node, X266HIP_EINVAL, "unknown node option");
}

int xHipNodeRcclInfo(int *version, char *path, size_t path_cap)
{
    Rccl *R = rccl();
    if (!R) return X266HIP_ECOMM;
    if (version) {
        int v = 0;
        if (R->GetVersion) (void)R->GetVersion(&v);
        *version = v;
    }
    if (path && path_cap) {
        Dl_info info;
        const char *p = dladdr((void *)R->GroupStart, &info) && info.dli_fname ? info.dli_fname : "";
        std::snprintf(path, path_cap, "%s", p);
    }
    return X266HIP_OK;
}

static int self_test_impl(x266hip_node *node);

int xHipNodeSelfTest(x266hip_node *node)
{
    if (!node) return X266HIP_EINVAL;
    if (node->failed) return nfail(node, X266HIP_ECOMM, "an earlier step of this node failed and its communicators were aborted");
    if (!node->have_rccl) return nfail(node, X266HIP_ECOMM, "RCCL did not initialise");
    const int rc = self_test_impl(node);
    if (rc == X266HIP_ECOMM || rc == X266HIP_EDEVICE) abort_comms(node);   // a rank may sit in a group that will never complete
    return rc;
}

static int self_test_impl(x266hip_node *node)
{
    Rccl *R = rccl();
    const size_t n = 1 << 16;                                         // 64 Ki uint32 each way
    const int W = node->world;
    std::vector<uint32_t> host(n);
    for (LocalRank &lr : node->local) {
        DeviceScope dev(lr.device);
        int rc = grow(node, lr.selftest, 3 * n * 4 + 16);
        if (rc) return rc;
        for (size_t i = 0; i < n; ++i) host[i] = (uint32_t)lr.rank * 0x9E3779B9u + (uint32_t)i * 2654435761u;
        N_HIP(node, hipMemcpyAsync(lr.selftest.p, host.data(), n * 4, hipMemcpyHostToDevice, lr.comm_stream));
        N_HIP(node, hipMemsetAsync((char *)lr.selftest.p + n * 4, 0, 2 * n * 4 + 16, lr.comm_stream));
        N_HIP(node, hipStreamSynchronize(lr.comm_stream));          // the host vector is reused for the next rank
    }
    N_NCCL(node, R->GroupStart());
    for (LocalRank &lr : node->local) {
        char *base = (char *)lr.selftest.p;
        ncclResult_t r = R->Send(base, n * 4, ncclUint8, (lr.rank + 1) % W, lr.comm, lr.comm_stream);
        if (r == ncclSuccess) r = R->Recv(base + n * 4, n * 4, ncclUint8, (lr.rank + W - 1) % W, lr.comm, lr.comm_stream);
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            return nfail(node, X266HIP_ECOMM, std::string("self-test send/recv: ") + R->GetErrorString(r));
        }
    }
    N_NCCL(node, R->GroupEnd());
    // all-reduce (sum, uint32 wraps) of what every rank RECEIVED: equals the sum of what every rank sent
    N_NCCL(node, R->GroupStart());
    for (LocalRank &lr : node->local) {
        char *base = (char *)lr.selftest.p;
        ncclResult_t r = R->AllReduce(base + n * 4, base + 2 * n * 4, n, ncclUint32, ncclSum, lr.comm, lr.comm_stream);
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            return nfail(node, X266HIP_ECOMM, std::string("self-test all-reduce: ") + R->GetErrorString(r));
        }
    }
    N_NCCL(node, R->GroupEnd());
    std::vector<uint32_t> got(2 * n);
    for (LocalRank &lr : node->local) {
        DeviceScope dev(lr.device);
        N_HIP(node, hipMemcpyAsync(got.data(), (char *)lr.selftest.p + n * 4, 2 * n * 4, hipMemcpyDeviceToHost, lr.comm_stream));
        N_HIP(node, hipStreamSynchronize(lr.comm_stream));
        const uint32_t from = (uint32_t)((lr.rank + W - 1) % W);
        for (size_t i = 0; i < n; ++i) {
            if (got[i] != from * 0x9E3779B9u + (uint32_t)i * 2654435761u)
                return nfail(node, X266HIP_ECOMM, "self-test: received data differs from what the previous rank sent");
            uint32_t sum = 0;
            for (int r = 0; r < W; ++r) sum += (uint32_t)r * 0x9E3779B9u + (uint32_t)i * 2654435761u;
            if (got[n + i] != sum) return nfail(node, X266HIP_ECOMM, "self-test: all-reduce result is wrong");
        }
    }
    return X266HIP_OK;
}

// ---- frame stream ---------------------------------------------------------------------------------
void xNodeStreamFree(x266hip_nstream *s)
{
    if (!s) return;
    for (size_t i = 0; i < s->per.size(); ++i) {
        LocalRank &lr = s->node->local[i];
        DeviceScope dev(lr.device);
        (void)hipStreamSynchronize(lr.comm_stream);
        for (hipStream_t cs : lr.compute) (void)hipStreamSynchronize(cs);
        x266hip_nstream::PerRank &p = s->per[i];
        for (int sl = 0; sl < x266hip_nstream::kSlots; ++sl) {
            for (int l = 0; l < 4; ++l) {
                if (p.in[sl][l]) (void)hipFree(p.in[sl][l]);
                if (p.out[sl][l]) (void)hipFree(p.out[sl][l]);
            }
            if (p.ev_xfer[sl]) (void)hipEventDestroy(p.ev_xfer[sl]);
            if (p.ev_done[sl]) (void)hipEventDestroy(p.ev_done[sl]);
        }
        if (p.ev_producer) (void)hipEventDestroy(p.ev_producer);
    }
    delete s;
}

int xNodeStreamCreate(x266hip_node *node, int n_lanes, const int *ops, const size_t *max_units, x266hip_nstream **out)
{
    if (!node || !out || !ops || !max_units || n_lanes < 1 || n_lanes > 4) return X266HIP_EINVAL;
    *out = nullptr;
    for (int l = 0; l < n_lanes; ++l)
        if (ops[l] < 0 || ops[l] > 2) return nfail(node, X266HIP_EINVAL, "xNodeStreamCreate: op must be 0, 1 or 2");
    x266hip_nstream *s = new (std::nothrow) x266hip_nstream;
    if (!s) return X266HIP_ENOMEM;
    s->node = node;
    s->n_lanes = n_lanes;
    for (int l = 0; l < n_lanes; ++l) {
        s->op[l] = ops[l];
        s->max_units[l] = max_units[l];
    }
    s->per.resize(node->local.size());
    for (size_t i = 0; i < node->local.size(); ++i) {
        LocalRank &lr = node->local[i];
        DeviceScope dev(lr.device);
        x266hip_nstream::PerRank &p = s->per[i];
        bool ok = true;
        for (int sl = 0; sl < x266hip_nstream::kSlots && ok; ++sl) {
            ok = hipEventCreateWithFlags(&p.ev_xfer[sl], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&p.ev_done[sl], hipEventDisableTiming) == hipSuccess;
            if (lr.rank == node->root) continue;                      // the root works in the caller's buffers
            for (int l = 0; l < n_lanes && ok; ++l) {
                const size_t cap = (max_units[l] + (size_t)node->world - 1) / (size_t)node->world;   // largest shard
                if (cap == 0) continue;
                ok = hipMalloc(&p.in[sl][l], cap * kInUnit[ops[l]]) == hipSuccess &&
                     hipMalloc(&p.out[sl][l], cap * kOutUnit[ops[l]]) == hipSuccess;
            }
        }
        if (ok) ok = hipEventCreateWithFlags(&p.ev_producer, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            xNodeStreamFree(s);
            return nfail(node, X266HIP_ENOMEM, "xNodeStreamCreate: device allocation failed");
        }
    }
    *out = s;
    return X266HIP_OK;
}

int xNodeFrameStreamCreate(x266hip_node *node, int width, int height, x266hip_nstream **s)
{
    if (!node || width < 32 || height < 32 || (width & 31) || (height & 31))
        return nfail(node, X266HIP_EINVAL, "xNodeFrameStreamCreate: width and height must be multiples of 32");
    const int ops[2] = {0, 2};
    const size_t units[2] = {(size_t)(width / 32) * (size_t)(height / 32), (size_t)(width / 8) * (size_t)(height / 8)};
    return xNodeStreamCreate(node, 2, ops, units, s);
}

}  // extern "C"

namespace {

// Step t: transfers {inputs of frame t, results of frame t-2}, then kernels of frame t.
int stream_step_impl(x266hip_nstream *s, const void *const *d_in, void *const *d_out, const size_t *units, void *producer_stream, bool has_frame, bool *posted,
                     bool *root_only_refusal)
{
    x266hip_node *node = s->node;
    if (node->failed) return nfail(node, X266HIP_ECOMM, "an earlier step of this node failed and its communicators were aborted");
    const int W = node->world, root = node->root;
    const long t = s->n_steps;
    const int slot = (int)(t % x266hip_nstream::kSlots);
    x266hip_nstream::Step &cur = s->ring[t % x266hip_nstream::kRing];
    const bool drives_root = root_rank(node) != nullptr;
    cur = x266hip_nstream::Step();
    cur.has_frame = has_frame;
    if (has_frame) {
        for (int l = 0; l < s->n_lanes; ++l) {
            cur.units[l] = units ? units[l] : s->max_units[l];
            if (cur.units[l] > s->max_units[l]) return nfail(node, X266HIP_EINVAL, "xNodeStreamPush: units exceed the stream's max_units");
            if (drives_root && cur.units[l]) {
                if (!d_in || !d_out || !d_in[l] || !d_out[l] || (((uintptr_t)d_in[l] | (uintptr_t)d_out[l]) & 15u)) {
                    *root_only_refusal = true;                         // the peers cannot see this check: they post step t regardless
                    return nfail(node, X266HIP_EINVAL, "xNodeStreamPush: NULL or unaligned frame buffer on the root");
                }
                cur.d_in[l] = (const char *)d_in[l];
                cur.d_out[l] = (char *)d_out[l];
            }
        }
        // buffer ownership (include/x266hip.h, xNodeStreamPush): this frame's outputs must not overlap the outputs of the frames that
        // are still in flight -- the previous X266_STREAM_OUT_RING - 1 steps unless a flush has completed them
        if (drives_root) {
            for (long back = 1; back < X266_STREAM_OUT_RING && back <= t - s->flushed_steps; ++back) {
                const x266hip_nstream::Step &prev = s->ring[(t - back) % x266hip_nstream::kRing];
                if (!prev.has_frame) continue;
                for (int l = 0; l < s->n_lanes; ++l)
                    for (int m = 0; m < s->n_lanes; ++m) {
                        if (!cur.units[l] || !prev.units[m]) continue;
                        const char *a0 = cur.d_out[l], *a1 = a0 + cur.units[l] * kOutUnit[s->op[l]];
                        const char *b0 = prev.d_out[m], *b1 = b0 + prev.units[m] * kOutUnit[s->op[m]];
                        if (a0 < b1 && b0 < a1) {
                            *root_only_refusal = true;
                            return nfail(node, X266HIP_EINVAL, "xNodeStreamPush: an output buffer overlaps the output of a frame still in flight (output rings need X266_STREAM_OUT_RING buffers)");
                        }
                    }
            }
        }
    }
    const x266hip_nstream::Step *old = t >= 2 ? &s->ring[(t - 2) % x266hip_nstream::kRing] : nullptr;
    if (old && !old->has_frame) old = nullptr;
    const int old_slot = (int)((t + x266hip_nstream::kSlots - 2) % x266hip_nstream::kSlots);   // frame t-2's slot

    // (1) the slot's previous frame (t-3) must have finished its kernels: its input slot is about to be
    //     overwritten (its output slot was sent at step t-1).  Host wait = bounded run-ahead (three frames).
    for (size_t i = 0; i < node->local.size(); ++i) {
        LocalRank &lr = node->local[i];
        x266hip_nstream::PerRank &p = s->per[i];
        DeviceScope dev(lr.device);
        if (p.done_recorded[slot]) {
            N_HIP(node, hipEventSynchronize(p.ev_done[slot]));
            if (W > 1) N_HIP(node, hipStreamWaitEvent(lr.comm_stream, p.ev_done[slot], 0));
        }
        if (W > 1 && old) {                                            // frame t-2's results leave in this step: behind its kernels (device-side wait)
            if (p.done_recorded[old_slot]) N_HIP(node, hipStreamWaitEvent(lr.comm_stream, p.ev_done[old_slot], 0));
        }
        // ... and the transfers of the step that last used this slot (step t-3's group on the communication stream) must have
        // completed: with one process per GPU nothing else keeps the root from running many groups ahead of lagging peers, and
        // both the buffer-ownership rule of xNodeStreamPush (inputs reusable after three later steps) and xNodeStreamWait rest on it
        if (p.xfer_recorded[slot]) N_HIP(node, hipEventSynchronize(p.ev_xfer[slot]));
        if (lr.rank == root && has_frame) {                           // inputs come from the caller's stream
            hipStream_t cs = lr.compute[slot];
            const bool in_order = W == 1 && producer_stream && (hipStream_t)producer_stream == cs;   // produced on the slot's own stream: stream order suffices
            if (!in_order) {
                N_HIP(node, hipEventRecord(p.ev_producer, (hipStream_t)producer_stream));
                if (W > 1) N_HIP(node, hipStreamWaitEvent(lr.comm_stream, p.ev_producer, 0));
                N_HIP(node, hipStreamWaitEvent(cs, p.ev_producer, 0));
            }
        }
    }
    if (node->transport == 1 && has_frame && drives_root) {           // peer copies run on the PEERS' streams and read the root's frame
        hipEvent_t prod = nullptr;
        for (size_t i = 0; i < node->local.size(); ++i)
            if (node->local[i].rank == root) prod = s->per[i].ev_producer;
        for (LocalRank &lr : node->local) {
            if (lr.rank == root) continue;
            DeviceScope dev(lr.device);
            N_HIP(node, hipStreamWaitEvent(lr.comm_stream, prod, 0));
        }
    }
    // (2) the step's transfers, same lane order on both ends of every (root, peer) pair
    std::vector<Xfer> xs;
    for (size_t i = 0; i < node->local.size(); ++i) {
        LocalRank &lr = node->local[i];
        x266hip_nstream::PerRank &p = s->per[i];
        for (int phase = 0; phase < 2; ++phase) {                     // 0: inputs of frame t, 1: results of frame t-2
            const x266hip_nstream::Step *st = phase == 0 ? (has_frame ? &cur : nullptr) : old;
            if (!st) continue;
            for (int l = 0; l < s->n_lanes; ++l) {
                const size_t unit = phase == 0 ? kInUnit[s->op[l]] : kOutUnit[s->op[l]];
                if (lr.rank == root) {
                    for (int peer = 0; peer < W; ++peer) {
                        if (peer == root) continue;
                        size_t b, e;
                        shard(st->units[l], peer, W, &b, &e);
                        if (e == b) continue;
                        char *buf = phase == 0 ? const_cast<char *>(st->d_in[l]) + b * unit : st->d_out[l] + b * unit;
                        xs.push_back({(int)i, phase == 0, peer, buf, (e - b) * unit});
                    }
                } else {
                    size_t b, e;
                    shard(st->units[l], lr.rank, W, &b, &e);
                    if (e == b) continue;
                    xs.push_back({(int)i, phase == 1, root, phase == 0 ? p.in[slot][l] : p.out[old_slot][l], (e - b) * unit});
                }
            }
        }
    }
    *posted = true;                                                  // from here on a failure leaves peers in a group: the caller aborts
    int rc = post_transfers(node, xs);
    if (rc) return rc;
    // (3) kernels of frame t behind the transfers
    for (size_t i = 0; i < node->local.size(); ++i) {
        LocalRank &lr = node->local[i];
        x266hip_nstream::PerRank &p = s->per[i];
        DeviceScope dev(lr.device);
        if (W > 1) {                                                    // one rank: nothing travels, the communication stream stays idle
            N_HIP(node, hipEventRecord(p.ev_xfer[slot], lr.comm_stream));
            p.xfer_recorded[slot] = true;
        }
        if (!has_frame) continue;
        hipStream_t cs = lr.compute[slot];   // the slot's own stream: frame t-2 (same buffers) is ahead of frame t on it
        if (lr.rank != root) N_HIP(node, hipStreamWaitEvent(cs, p.ev_xfer[slot], 0));   // the root works in place: nothing to wait for
        const void *lane_in[4] = {};
        void *lane_out[4] = {};
        size_t lane_n[4] = {};
        for (int l = 0; l < s->n_lanes; ++l) {
            size_t b, e;
            shard(cur.units[l], lr.rank, W, &b, &e);
            lane_in[l] = lr.rank == root ? (const void *)(cur.d_in[l] + b * kInUnit[s->op[l]]) : p.in[slot][l];
            lane_out[l] = lr.rank == root ? (void *)(cur.d_out[l] + b * kOutUnit[s->op[l]]) : p.out[slot][l];
            lane_n[l] = e - b;
        }
        if (s->n_lanes == 2 && s->op[0] == 0 && s->op[1] == 2 && node->fused_frame_lanes) {
            // the frame stream's two lanes (DCT32 forward + SATD): one grid instead of two submissions
            N_X(node, lr, xDct32SatdFrameDev(lr.ctx, (const int16_t *)lane_in[0], (int16_t *)lane_out[0], lane_n[0],
                                             (const int16_t *)lane_in[1], (uint32_t *)lane_out[1], lane_n[1], cs));
        } else {
            for (int l = 0; l < s->n_lanes; ++l) {
                rc = launch(node, lr, s->op[l], lane_in[l], lane_out[l], lane_n[l], cs);
                if (rc) return rc;
            }
        }
        N_HIP(node, hipEventRecord(p.ev_done[slot], cs));
        p.done_recorded[slot] = true;
    }
    s->n_steps = t + 1;
    return X266HIP_OK;
}

int stream_step(x266hip_nstream *s, const void *const *d_in, void *const *d_out, const size_t *units, void *producer_stream, bool has_frame)
{
    bool posted = false, root_only_refusal = false;
    const int rc = stream_step_impl(s, d_in, d_out, units, producer_stream, has_frame, &posted, &root_only_refusal);
    // When does a failed step cost the node?  (a) after the step's group was posted, whatever the code (a per-rank launch can still refuse
    // an argument there), and (b) on any non-argument failure before it: the other ranks will post a group this rank never joins.
    // (c) With one process per GPU also on the argument checks that only the ROOT can make (NULL / unaligned frame buffer, an output
    // still in flight): the peers post step t regardless, so this rank's communicators are aborted and the node marked
    // failed -- every later call says ECOMM instead of pairing step t+1 with the peers' step t, and the peers learn of it the way
    // they learn of any lost rank.  Argument errors EVERY rank sees identically and that post nothing ("units exceed max_units":
    // all ranks pass the same counts) stay harmless: the call returns EINVAL and the stream is as it was.
    const bool peers_elsewhere = !s->node->single_process && s->node->world > 1;
    if (rc != X266HIP_OK && (posted || rc != X266HIP_EINVAL || (peers_elsewhere && root_only_refusal))) abort_comms(s->node);
    return rc;
}

}  // namespace

extern "C" {

int xNodeStreamPush(x266hip_nstream *s, const void *const *d_in, void *const *d_out, const size_t *units,
                    void *producer_stream, long *ticket)
{
    if (!s) return X266HIP_EINVAL;
    if (ticket) *ticket = s->n_steps;
    return stream_step(s, d_in, d_out, units, producer_stream, true);
}

void *xNodeStreamNextSlotStream(x266hip_nstream *s)
{
    if (!s) return nullptr;
    LocalRank *r = root_rank(s->node);
    if (!r) return nullptr;
    return (void *)r->compute[s->n_steps % x266hip_nstream::kSlots];
}

int xNodeStreamFlush(x266hip_nstream *s)
{
    if (!s) return X266HIP_EINVAL;
    for (int i = 0; i < 2; ++i) {
        int rc = stream_step(s, nullptr, nullptr, nullptr, nullptr, false);
        if (rc) return rc;
    }
    for (LocalRank &lr : s->node->local) {
        DeviceScope dev(lr.device);
        N_HIP(s->node, hipStreamSynchronize(lr.comm_stream));
        for (hipStream_t cs : lr.compute) N_HIP(s->node, hipStreamSynchronize(cs));
    }
    s->flushed_steps = s->n_steps;
    return X266HIP_OK;
}

int xNodeStreamWait(x266hip_nstream *s, long ticket)
{
    if (!s || ticket < 0) return X266HIP_EINVAL;
    if (ticket + 2 >= s->n_steps) return nfail(s->node, X266HIP_EINVAL, "xNodeStreamWait: the step's results travel two steps later (Push or Flush first)");
    constexpr long K = x266hip_nstream::kSlots;
    // the frame's kernels: their event is host-synchronised anyway when the slot comes round again (step ticket + K)
    if (s->n_steps <= ticket + K) {
        const int slot = (int)(ticket % K);
        for (size_t i = 0; i < s->node->local.size(); ++i) {
            DeviceScope dev(s->node->local[i].device);
            if (s->per[i].done_recorded[slot]) N_HIP(s->node, hipEventSynchronize(s->per[i].ev_done[slot]));
        }
    }
    // its results' way back: step ticket + 2's transfers (likewise synchronised at step ticket + 2 + K)
    if (s->n_steps <= ticket + 2 + K) {
        const int slot = (int)((ticket + 2) % K);
        for (size_t i = 0; i < s->node->local.size(); ++i) {
            DeviceScope dev(s->node->local[i].device);
            if (s->per[i].xfer_recorded[slot]) N_HIP(s->node, hipEventSynchronize(s->per[i].ev_xfer[slot]));
        }
    }
    if (ticket + 1 > s->flushed_steps) s->flushed_steps = ticket + 1;   // results return in step order: everything up to the ticket is complete
    return X266HIP_OK;
}

int xNodeBatchScatterGather(x266hip_node *node, int op, const void *d_in, void *d_out, size_t n_units, size_t chunk_units)
{
    if (!node || op < 0 || op > 2) return X266HIP_EINVAL;
    if (n_units == 0) return X266HIP_OK;
    if (chunk_units == 0) {
        // one rank: nothing travels, so nothing to pipeline -- one launch.  Otherwise 8 MiB of input per rank and chunk
        // (about 50 us on an xGMI link, well above a group's launch cost), but at least four chunks to overlap
        const size_t floor_units = (size_t)(8u << 20) / kInUnit[op];
        chunk_units = node->world == 1 ? n_units : floor_units * (size_t)node->world;
        if (node->world > 1 && chunk_units > n_units / 4) chunk_units = n_units / 4 > floor_units ? n_units / 4 : floor_units;
    }
    if (chunk_units > n_units) chunk_units = n_units;
    // one internal stream per op, kept on the node (grow-only): creating one per call means slot allocations, events and
    // -- on release -- hipFree's device synchronisation inside what callers time
    int rc = X266HIP_OK;
    if (!node->sg_stream[op] || node->sg_cap[op] < chunk_units) {
        if (node->sg_stream[op]) { xNodeStreamFree(node->sg_stream[op]); node->sg_stream[op] = nullptr; }
        rc = xNodeStreamCreate(node, 1, &op, &chunk_units, &node->sg_stream[op]);
        if (rc) return rc;
        node->sg_cap[op] = chunk_units;
    }
    x266hip_nstream *s = node->sg_stream[op];
    const bool drives_root = root_rank(node) != nullptr;
    for (size_t done = 0; done < n_units && rc == X266HIP_OK; done += chunk_units) {
        const size_t cnt = n_units - done < chunk_units ? n_units - done : chunk_units;
        const void *in = drives_root ? (const void *)((const char *)d_in + done * kInUnit[op]) : nullptr;
        void *out = drives_root ? (void *)((char *)d_out + done * kOutUnit[op]) : nullptr;
        rc = xNodeStreamPush(s, &in, &out, &cnt, nullptr, nullptr);
    }
    if (rc == X266HIP_OK) rc = xNodeStreamFlush(s);
    return rc;
}

static int node_search_impl(x266hip_node *node, const uint8_t *d_cur, intptr_t cur_stride, const uint8_t *d_ref,
                            intptr_t ref_stride, int width, int height, int range, int n_stripes, x266_me_result_t *d_best);

int xNodeSatd8x8Search(x266hip_node *node, const uint8_t *d_cur, intptr_t cur_stride, const uint8_t *d_ref,
                       intptr_t ref_stride, int width, int height, int range, int n_stripes, x266_me_result_t *d_best)
{
    if (!node) return X266HIP_EINVAL;
    if (node->failed) return nfail(node, X266HIP_ECOMM, "an earlier step of this node failed and its communicators were aborted");
    // argument errors are detected before anything is posted (node_search_impl validates first); everything later leaves peers in a group
    const int rc = node_search_impl(node, d_cur, cur_stride, d_ref, ref_stride, width, height, range, n_stripes, d_best);
    if (rc == X266HIP_ECOMM || rc == X266HIP_EDEVICE || rc == X266HIP_ENOMEM) abort_comms(node);
    return rc;
}

static int node_search_impl(x266hip_node *node, const uint8_t *d_cur, intptr_t cur_stride, const uint8_t *d_ref,
                            intptr_t ref_stride, int width, int height, int range, int n_stripes, x266_me_result_t *d_best)
{
    if (width < 8 || height < 8 || (width & 7) || (height & 7)) return nfail(node, X266HIP_EINVAL, "xNodeSatd8x8Search: frame size must be a multiple of 8");
    if (range < 1 || range > 64) return nfail(node, X266HIP_EINVAL, "xNodeSatd8x8Search: range must be 1..64");
    if (cur_stride < width || ref_stride < width + 2 * range) return nfail(node, X266HIP_EINVAL, "xNodeSatd8x8Search: stride too small");
    const int W = node->world, root = node->root;
    if (n_stripes <= 0) n_stripes = W;
    LocalRank *rl = root_rank(node);
    if (rl && (!d_cur || !d_ref || !d_best || ((uintptr_t)d_best & 7u))) return nfail(node, X266HIP_EINVAL, "xNodeSatd8x8Search: NULL or unaligned buffer on the root");
    const size_t bw = (size_t)(width / 8);

    struct Stripe { int b, e, r0, r1; size_t cur_bytes, ref_bytes, best_bytes; int owner; };
    std::vector<Stripe> stripes((size_t)n_stripes);
    for (int r = 0; r < W; ++r) {
        size_t sb, se;
        shard((size_t)n_stripes, r, W, &sb, &se);                    // contiguous runs of stripes per rank
        for (size_t si = sb; si < se; ++si) {
            Stripe &st = stripes[si];
            (void)xMeStripePlan(height, range, (int)si, n_stripes, &st.b, &st.e, &st.r0, &st.r1);
            st.owner = r;
            const size_t rows = (size_t)(st.e - st.b) * 8;
            st.cur_bytes = rows ? (rows - 1) * (size_t)cur_stride + (size_t)width : 0;          // rows travel with their stride
            st.ref_bytes = rows ? (size_t)(st.r1 - st.r0 - 1) * (size_t)ref_stride + (size_t)(width + 2 * range) : 0;
            st.best_bytes = (size_t)(st.e - st.b) * bw * sizeof(x266_me_result_t);
        }
    }
    // stripe buffers on the owners (the root's own stripes are searched in place unless me_local_copy)
    std::vector<Xfer> in_x, out_x;
    struct Work { size_t local; const uint8_t *cur, *ref_origin; x266_me_result_t *best; int rows; };
    struct CopyBack { size_t local; const void *src; x266_me_result_t *dst; size_t bytes; };
    std::vector<Work> work;
    std::vector<CopyBack> copy_back;
    for (size_t i = 0; i < node->local.size(); ++i) {
        LocalRank &lr = node->local[i];
        DeviceScope dev(lr.device);
        size_t k = 0;
        for (size_t si = 0; si < stripes.size(); ++si) {
            const Stripe &st = stripes[si];
            if (st.e == st.b) continue;
            const uint8_t *src_cur = d_cur ? d_cur + (intptr_t)st.b * 8 * cur_stride : nullptr;
            const uint8_t *src_ref = d_ref ? d_ref + (intptr_t)st.r0 * ref_stride - range : nullptr;   // first byte of the padded row r0
            x266_me_result_t *dst_best = d_best ? d_best + (size_t)st.b * bw : nullptr;
            if (lr.rank == root && st.owner != root) {
                in_x.push_back({(int)i, true, st.owner, const_cast<uint8_t *>(src_cur), st.cur_bytes});
                in_x.push_back({(int)i, true, st.owner, const_cast<uint8_t *>(src_ref), st.ref_bytes});
                out_x.push_back({(int)i, false, st.owner, dst_best, st.best_bytes});
            }
            if (st.owner != lr.rank) continue;
            if (lr.rank == root && !node->me_local_copy) {
                work.push_back({i, src_cur, src_ref + (intptr_t)range * ref_stride + range, dst_best, (st.e - st.b) * 8});
                continue;
            }
            if (lr.me_bufs.size() < 3 * (k + 1)) lr.me_bufs.resize(3 * (k + 1));
            DevBuf &bc = lr.me_bufs[3 * k], &br = lr.me_bufs[3 * k + 1], &bb = lr.me_bufs[3 * k + 2];
            ++k;
            int rc = grow(node, bc, st.cur_bytes + 16);
            if (!rc) rc = grow(node, br, st.ref_bytes + 16);
            if (!rc) rc = grow(node, bb, st.best_bytes + 16);
            if (rc) return rc;
            if (lr.rank == root) {                                    // me_local_copy: what a peer would receive, by device copies
                N_HIP(node, hipMemcpyAsync(bc.p, src_cur, st.cur_bytes, hipMemcpyDeviceToDevice, lr.comm_stream));
                N_HIP(node, hipMemcpyAsync(br.p, src_ref, st.ref_bytes, hipMemcpyDeviceToDevice, lr.comm_stream));
            } else {
                in_x.push_back({(int)i, false, root, bc.p, st.cur_bytes});
                in_x.push_back({(int)i, false, root, br.p, st.ref_bytes});
                out_x.push_back({(int)i, true, root, bb.p, st.best_bytes});
            }
            work.push_back({i, (const uint8_t *)bc.p, (const uint8_t *)br.p + (intptr_t)range * ref_stride + range,
                            (x266_me_result_t *)bb.p, (st.e - st.b) * 8});
            if (lr.rank == root) copy_back.push_back({i, bb.p, dst_best, st.best_bytes});
        }
    }
    // the caller's default stream produced the frame
    if (rl) {
        DeviceScope dev(rl->device);
        N_HIP(node, hipStreamSynchronize(nullptr));
    }
    int rc = post_transfers(node, in_x);
    if (rc) return rc;
    for (LocalRank &lr : node->local) {
        DeviceScope dev(lr.device);
        N_HIP(node, hipStreamSynchronize(lr.comm_stream));           // one-shot call: plain host ordering
    }
    for (const Work &w : work) {
        LocalRank &lr = node->local[w.local];
        N_X(node, lr, xSatd8x8SearchDev(lr.ctx, w.cur, cur_stride, w.ref_origin, ref_stride, width, w.rows, range, w.best, nullptr, lr.compute[0]));
    }
    for (LocalRank &lr : node->local) {
        DeviceScope dev(lr.device);
        N_HIP(node, hipStreamSynchronize(lr.compute[0]));
    }
    for (const CopyBack &c : copy_back) {                             // the root's own copied stripes
        LocalRank &lr = node->local[c.local];
        DeviceScope dev(lr.device);
        N_HIP(node, hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToDevice, lr.comm_stream));
    }
    rc = post_transfers(node, out_x);
    if (rc) return rc;
    for (LocalRank &lr : node->local) {
        DeviceScope dev(lr.device);
        N_HIP(node, hipStreamSynchronize(lr.comm_stream));
    }
    return X266HIP_OK;
}

}  // extern "C"
