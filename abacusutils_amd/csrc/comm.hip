// RCCL communicator behind the C ABI (include/abacus_hip.h, "multi-GPU communicator"): one process per GPU, the
// collectives of the slab-decomposed P(k), the sharded HOD and the slab pair counts without any Python
// framework in between.  librccl is opened lazily (dlopen) the first time a communicator is created, so single-GPU users
// never load it.
//
//   rendezvous     rank 0 calls abacus_comm_unique_id and hands the 128 bytes to the other ranks out of band (the Python
//                  side uses a file or TCP rendezvous, abacusutils_amd/comm.py); every rank then calls abacus_comm_init
//                  (ncclCommInitRank on the device the library was bound to with abacus_set_device).
//   data path      all-to-all (equal blocks and all-to-all-v) as ONE group of ncclSend / ncclRecv per peer - xGMI is
//                  point to point, so every link carries its 1/W of the slab at the same time (a ring formulation
//                  would be bound by one link) -, ring exchange of ghost blocks with the two neighbours, all-reduce.
//   streams        collectives are enqueued on the library stream (ordered with the kernels around them, no host
//                  synchronisation); `async` forms run on the communicator's own stream behind an event fork, so a
//                  chunk of the pencil transpose is on the links while the next chunk's FFT passes run
//                  (abacus_comm_join puts the library stream behind them again).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "../../include/abacus_hip.h"
#include "common.hpp"

using namespace abacus;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
} R;

int load_rccl() {
    if (R.handle) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail("RCCL not found (librccl.so.1): %s", dlerror());
#define SYM(field, name)                                                        \
    *(void **)(&R.field) = dlsym(h, name);                                      \
    if (!R.field) return fail("RCCL symbol %s missing: %s", name, dlerror());
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommAbort, "ncclCommAbort")
    SYM(CommGetAsyncError, "ncclCommGetAsyncError")
    SYM(GetErrorString, "ncclGetErrorString")
    SYM(GetVersion, "ncclGetVersion")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(AllReduce, "ncclAllReduce")
    SYM(AllGather, "ncclAllGather")
#undef SYM
    R.handle = h;
    return 0;
}

#define NCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) return ::abacus::fail("%s: %s (%s:%d)", #expr, R.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

}  // namespace

struct abacus_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1;
    hipStream_t side = nullptr;      // the communicator's own stream (async forms)
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    bool side_busy = false;
    DevBuf scratch;                  // staging of the host-buffer conveniences
    uint64_t bytes_sent = 0;         // to other ranks, data-path collectives only (bench: GB/s per link)
};

namespace {

int dtype_of(int code, ncclDataType_t *t, size_t *size) {
    switch (code) {
        case 0: *t = ncclInt64, *size = 8; return 0;
        case 1: *t = ncclFloat64, *size = 8; return 0;
        case 2: *t = ncclFloat32, *size = 4; return 0;
        case 3: *t = ncclUint64, *size = 8; return 0;
    }
    return fail("abacus_comm: unknown dtype code %d (0 int64, 1 float64, 2 float32, 3 uint64)", code);
}

hipStream_t pick_stream(abacus_comm *c, int async) {
    if (!async) return stream();
    // fork: the side stream continues behind everything enqueued on the library stream so far (every async operation
    // forks again: it has to see the kernels enqueued since the previous one)
    (void)hipEventRecord(c->fork_ev, stream());
    (void)hipStreamWaitEvent(c->side, c->fork_ev, 0);
    c->side_busy = true;
    return c->side;
}

// one group of sends / receives: block p of `send` (at soff[p], sbytes[p]) goes to rank p, block p of `recv` comes from it
int exchange(abacus_comm *c, const char *send, const uint64_t *sbytes, const uint64_t *soff, char *recv, const uint64_t *rbytes,
             const uint64_t *roff, hipStream_t s) {
    const int W = c->world, me = c->rank;
    if (sbytes[me] != rbytes[me]) return fail("abacus_comm: a rank's block to itself must have the same size on both sides");
    if (sbytes[me]) HIP_TRY(hipMemcpyAsync(recv + roff[me], send + soff[me], sbytes[me], hipMemcpyDeviceToDevice, s));
    if (W == 1) return 0;
    NCCL_TRY(R.GroupStart());
    // a failing Send / Recv must not leave the group open: every later RCCL call of this thread (the barrier or all-reduce
    // that would report the failure included) would queue into it and never run.  Keep the first error, always close.
    ncclResult_t bad = ncclSuccess;
    const char *what = "";
    for (int d = 1; d < W && bad == ncclSuccess; d++) {   // peers in rotated order: rank r talks to r+d / r-d, every link pair busy in every round
        const int to = (me + d) % W, from = (me - d + W) % W;
        if (sbytes[to] && (bad = R.Send(send + soff[to], sbytes[to], ncclUint8, to, c->nccl, s)) != ncclSuccess) what = "ncclSend";
        if (bad == ncclSuccess && rbytes[from] &&
            (bad = R.Recv(recv + roff[from], rbytes[from], ncclUint8, from, c->nccl, s)) != ncclSuccess)
            what = "ncclRecv";
        if (bad == ncclSuccess) c->bytes_sent += sbytes[to];
    }
    const ncclResult_t endr = R.GroupEnd();
    if (bad != ncclSuccess) return fail("abacus_comm: %s: %s", what, R.GetErrorString(bad));
    if (endr != ncclSuccess) return fail("abacus_comm: ncclGroupEnd: %s", R.GetErrorString(endr));
    return 0;
}

}  // namespace

extern "C" {

int abacus_comm_unique_id(void *id, int len) {
    ABACUS_ENTER();
    if (!id || len < (int)sizeof(ncclUniqueId)) return fail("abacus_comm_unique_id: the buffer must hold %d bytes", (int)sizeof(ncclUniqueId));
    ABACUS_TRY(load_rccl());
    ncclUniqueId u;
    NCCL_TRY(R.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return 0;
}

int abacus_comm_init(int rank, int world, const void *id, int len, abacus_comm **out) {
    auto *c = new abacus_comm();
    ncclUniqueId u;
    {
        ABACUS_ENTER();   // binds the device chosen with abacus_set_device: ncclCommInitRank takes the current device
        if (!out || world < 1 || rank < 0 || rank >= world || !id || len < (int)sizeof(ncclUniqueId)) {
            delete c;
            return !out ? fail("abacus_comm_init: null output")
                        : (!id || len < (int)sizeof(ncclUniqueId)) ? fail("abacus_comm_init: the unique id must hold %d bytes", (int)sizeof(ncclUniqueId))
                                                                    : fail("abacus_comm_init: rank %d of %d", rank, world);
        }
        const int rc = load_rccl();
        if (rc) {
            delete c;
            return rc;
        }
        c->rank = rank, c->world = world;
        memcpy(&u, id, sizeof u);
    }
    // ncclCommInitRank blocks until EVERY rank has joined.  The library's API lock is NOT held across it, so that a caller whose
    // deadline on the join expires (comm.py: RcclJoinTimeout from a helper thread) is not queued behind a dead call while it
    // reports the failure.  Policy (comm.py, bench.py): such a process makes NO further library call and ends at once
    // (os._exit) - the helper thread may still return into the code below - and the launcher starts a fresh process; the
    // file-barrier fallback of the HOD leg is only taken when the join FAILED, never when it timed out.
    const ncclResult_t r = R.CommInitRank(&c->nccl, world, u, rank);
    ABACUS_ENTER();
    if (r != ncclSuccess) {
        delete c;
        return fail("ncclCommInitRank(rank %d of %d): %s", rank, world, R.GetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming) != hipSuccess) {
        (void)abacus_comm_free(c);
        return fail("abacus_comm_init: stream / event creation failed");
    }
    *out = c;
    return 0;
}

int abacus_comm_info(const abacus_comm *c, int *rank, int *world, int *rccl_version, uint64_t *bytes_sent) {
    if (!c) return fail("abacus_comm_info: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (rccl_version) {
        *rccl_version = 0;
        if (R.GetVersion) (void)R.GetVersion(rccl_version);
    }
    if (bytes_sent) *bytes_sent = c->bytes_sent;
    return 0;
}

int abacus_comm_free(abacus_comm *c) {
    if (!c) return 0;
    std::lock_guard<std::recursive_mutex> guard(::abacus::api_mutex());
    (void)hipStreamSynchronize(stream());
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->nccl) (void)R.CommDestroy(c->nccl);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->join_ev) (void)hipEventDestroy(c->join_ev);
    (void)c->scratch.release();
    delete c;
    return 0;
}

int abacus_comm_abort(abacus_comm *c) {
    if (!c) return 0;
    if (c->nccl && R.CommAbort) (void)R.CommAbort(c->nccl);
    c->nccl = nullptr;
    return 0;
}

int abacus_comm_join(abacus_comm *c) {
    ABACUS_ENTER();
    if (!c) return fail("abacus_comm_join: null communicator");
    if (c->side_busy) {
        HIP_TRY(hipEventRecord(c->join_ev, c->side));
        HIP_TRY(hipStreamWaitEvent(stream(), c->join_ev, 0));
        c->side_busy = false;
    }
    return 0;
}

int abacus_comm_all_to_all(abacus_comm *c, const void *send, void *recv, uint64_t bytes_per_peer, int async) {
    ABACUS_ENTER();
    if (!c || !send || !recv) return fail("abacus_comm_all_to_all: null argument");
    if (send == recv) return fail("abacus_comm_all_to_all: in-place exchange is not supported");
    std::vector<uint64_t> n((size_t)c->world, bytes_per_peer), off((size_t)c->world);
    for (int p = 0; p < c->world; p++) off[p] = (uint64_t)p * bytes_per_peer;
    return exchange(c, (const char *)send, n.data(), off.data(), (char *)recv, n.data(), off.data(), pick_stream(c, async));
}

int abacus_comm_all_to_all_strided(abacus_comm *c, const void *send, void *recv, uint64_t peer_stride, uint64_t offset,
                                   uint64_t bytes, int async) {
    ABACUS_ENTER();
    if (!c || !send || !recv) return fail("abacus_comm_all_to_all_strided: null argument");
    if (offset + bytes > peer_stride) return fail("abacus_comm_all_to_all_strided: piece [%llu, +%llu) exceeds the peer block of %llu bytes",
                                                  (unsigned long long)offset, (unsigned long long)bytes, (unsigned long long)peer_stride);
    std::vector<uint64_t> n((size_t)c->world, bytes), off((size_t)c->world);
    for (int p = 0; p < c->world; p++) off[p] = (uint64_t)p * peer_stride + offset;
    return exchange(c, (const char *)send, n.data(), off.data(), (char *)recv, n.data(), off.data(), pick_stream(c, async));
}

int abacus_comm_all_to_all_v(abacus_comm *c, const void *send, const uint64_t *send_bytes, const uint64_t *send_off, void *recv,
                             const uint64_t *recv_bytes, const uint64_t *recv_off) {
    ABACUS_ENTER();
    if (!c || !send_bytes || !send_off || !recv_bytes || !recv_off) return fail("abacus_comm_all_to_all_v: null argument");
    return exchange(c, (const char *)send, send_bytes, send_off, (char *)recv, recv_bytes, recv_off, stream());
}

int abacus_comm_all_to_all_v_async(abacus_comm *c, const void *send, const uint64_t *send_bytes, const uint64_t *send_off, void *recv,
                                   const uint64_t *recv_bytes, const uint64_t *recv_off, int async) {
    ABACUS_ENTER();
    if (!c || !send_bytes || !send_off || !recv_bytes || !recv_off) return fail("abacus_comm_all_to_all_v_async: null argument");
    return exchange(c, (const char *)send, send_bytes, send_off, (char *)recv, recv_bytes, recv_off, pick_stream(c, async));
}

int abacus_comm_ring_exchange(abacus_comm *c, const void *to_left, const void *to_right, void *from_right, void *from_left,
                              uint64_t bytes) {
    ABACUS_ENTER();
    if (!c || !to_left || !to_right || !from_right || !from_left) return fail("abacus_comm_ring_exchange: null argument");
    const int W = c->world, left = (c->rank - 1 + W) % W, right = (c->rank + 1) % W;
    hipStream_t s = stream();
    if (W == 1) {   // periodic box on one rank: its own ghosts come back
        HIP_TRY(hipMemcpyAsync(from_right, to_left, bytes, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(from_left, to_right, bytes, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    // W == 2: both neighbours are the same rank.  Messages between one pair are matched in issue order: the peer's first
    // send is ITS to_left block, which is what arrives here from the right - so the receive order below holds for W = 2 too
    NCCL_TRY(R.GroupStart());
    ncclResult_t bad = R.Send(to_left, bytes, ncclUint8, left, c->nccl, s);       // first error kept, the group always closed
    if (bad == ncclSuccess) bad = R.Send(to_right, bytes, ncclUint8, right, c->nccl, s);
    if (bad == ncclSuccess) bad = R.Recv(from_right, bytes, ncclUint8, right, c->nccl, s);
    if (bad == ncclSuccess) bad = R.Recv(from_left, bytes, ncclUint8, left, c->nccl, s);
    const ncclResult_t endr = R.GroupEnd();
    if (bad != ncclSuccess) return fail("abacus_comm_ring_exchange: send / recv: %s", R.GetErrorString(bad));
    if (endr != ncclSuccess) return fail("abacus_comm_ring_exchange: ncclGroupEnd: %s", R.GetErrorString(endr));
    c->bytes_sent += 2 * bytes;
    return 0;
}

int abacus_comm_allreduce_dev(abacus_comm *c, void *buf, int64_t count, int dtype, int op) {
    ABACUS_ENTER();
    if (!c || !buf) return fail("abacus_comm_allreduce_dev: null argument");
    ncclDataType_t t;
    size_t sz;
    ABACUS_TRY(dtype_of(dtype, &t, &sz));
    if (op != 0 && op != 1) return fail("abacus_comm_allreduce: op must be 0 (sum) or 1 (max)");
    if (count <= 0) return 0;
    NCCL_TRY(R.AllReduce(buf, buf, (size_t)count, t, op == 0 ? ncclSum : ncclMax, c->nccl, stream()));
    return 0;
}

int abacus_comm_allreduce_host(abacus_comm *c, void *buf, int64_t count, int dtype, int op) {
    ABACUS_ENTER();
    if (!c || !buf) return fail("abacus_comm_allreduce_host: null argument");
    ncclDataType_t t;
    size_t sz;
    ABACUS_TRY(dtype_of(dtype, &t, &sz));
    if (count <= 0) return 0;
    ABACUS_TRY(c->scratch.reserve((size_t)count * sz));
    HIP_TRY(hipMemcpyAsync(c->scratch.p, buf, (size_t)count * sz, hipMemcpyHostToDevice, stream()));
    ABACUS_TRY(abacus_comm_allreduce_dev(c, c->scratch.p, count, dtype, op));
    HIP_TRY(hipMemcpyAsync(buf, c->scratch.p, (size_t)count * sz, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_comm_allgather_host(abacus_comm *c, const void *send, void *recv, uint64_t bytes) {
    ABACUS_ENTER();
    if (!c || !send || !recv) return fail("abacus_comm_allgather_host: null argument");
    if (bytes == 0) return 0;
    const size_t W = (size_t)c->world;
    ABACUS_TRY(c->scratch.reserve((W + 1) * bytes));
    char *d_send = c->scratch.as<char>(), *d_recv = d_send + bytes;
    HIP_TRY(hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, stream()));
    NCCL_TRY(R.AllGather(d_send, d_recv, bytes, ncclUint8, c->nccl, stream()));
    HIP_TRY(hipMemcpyAsync(recv, d_recv, W * bytes, hipMemcpyDeviceToHost, stream()));
    HIP_TRY(hipStreamSynchronize(stream()));
    return 0;
}

int abacus_comm_barrier(abacus_comm *c) {
    int64_t one = 1;
    ABACUS_TRY(abacus_comm_allreduce_host(c, &one, 1, 0, 0));
    if (one != c->world) return fail("abacus_comm_barrier: %lld of %d ranks arrived", (long long)one, c->world);
    return 0;
}

}  // extern "C"
