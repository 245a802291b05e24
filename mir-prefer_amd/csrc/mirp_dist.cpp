// Multi-GPU exchange step of the path behind the C-ABI: one process per GPU, contigs sharded over the ranks, and ONE collective on the data
// path -- the gather of the final loci list to one rank (the reference's analogue is `multiprocessing.Queue.put(list)` per piece,
// /root/reference/miR_PREFeR.py:2461-2499) -- plus the record exchange of the sharded SAM ingest (mirp_ingest.cpp).  RCCL (librccl.so.1 of the
// ROCm installation this library's HIP runtime comes from) is resolved with dlopen on first use: a single-GPU run never loads it, and the
// process holds exactly one HIP runtime and one RCCL instance (the host binding exchanges the 128-byte ncclUniqueId through whatever channel it
// has: a file, or a CPU-side store).
#include <dlfcn.h>
#include <time.h>
#include <cstdio>
#include <cstring>
#include <rccl/rccl.h>
#include "mirp_ctx.h"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    std::string err;
    bool load() {
        if (h) return true;
        // The RCCL that belongs to the HIP runtime THIS library is bound to: a process can hold two ROCm stacks with the same sonames (a Python
        // package that bundles its own libamdhip64.so.7 / librccl.so.1 next to the system installation; whichever was loaded first serves the
        // soname).  A collective must see the device pointers of the runtime that allocated them, so librccl is taken from the directory the
        // resolved hipMalloc lives in, by absolute path, before any by-name lookup.
        std::vector<std::string> names;
        Dl_info info;
        if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) { dir.resize(slash); names.push_back(dir + "/librccl.so.1"); names.push_back(dir + "/librccl.so"); }
        }
        names.push_back("librccl.so.1"); names.push_back("/opt/rocm/lib/librccl.so.1"); names.push_back("librccl.so");
        for (const std::string& name : names) {
            h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) { err = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
#define MIRP_SYM(field, sym)                                                                   \
    field = reinterpret_cast<decltype(field)>(dlsym(h, sym));                                  \
    if (!field) { err = std::string("librccl has no symbol ") + sym; dlclose(h); h = nullptr; return false; }
        MIRP_SYM(GetUniqueId, "ncclGetUniqueId") MIRP_SYM(CommInitRank, "ncclCommInitRank") MIRP_SYM(CommDestroy, "ncclCommDestroy")
        MIRP_SYM(AllGather, "ncclAllGather") MIRP_SYM(AllReduce, "ncclAllReduce") MIRP_SYM(Send, "ncclSend") MIRP_SYM(Recv, "ncclRecv")
        MIRP_SYM(GroupStart, "ncclGroupStart") MIRP_SYM(GroupEnd, "ncclGroupEnd") MIRP_SYM(GetErrorString, "ncclGetErrorString")
        MIRP_SYM(CommGetAsyncError, "ncclCommGetAsyncError") MIRP_SYM(CommAbort, "ncclCommAbort") MIRP_SYM(CommCount, "ncclCommCount")
        MIRP_SYM(CommCuDevice, "ncclCommCuDevice") MIRP_SYM(CommUserRank, "ncclCommUserRank")
#undef MIRP_SYM
        return true;
    }
};
Rccl g_rccl;

// Deadline of every wait on a peer (a collective's completion on the stream, a block of the local transport): MIRP_DIST_TIMEOUT_S seconds, default
// 600.  A rank that died (OOM kill, a failed allocation that made it leave before the collective, a lost GPU) never joins; without a deadline its
// peers would sit in hipStreamSynchronize for ever -- the reference's parent does exactly that on a crashed child.
double dist_timeout_s() {
    const char* e = std::getenv("MIRP_DIST_TIMEOUT_S");
    const double v = e ? std::atof(e) : 0.0;
    return v > 0.0 ? v : 600.0;
}
double mono_now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// ---- local transport: ranks that share one GPU (RCCL refuses two ranks on one device) exchange through files in a directory they all see.
// Every collective draws the next sequence number; rank s leaves its block for rank q in <dir>/x<seq>.<s>.<q> (written under a temporary
// name, then renamed) and polls for the blocks addressed to itself.  Host-staged, for tests and single-GPU debugging of a sharded run.
int local_exchange(mirp_ctx* c, const std::vector<std::pair<const void*, long long>>& send, std::vector<std::vector<char>>& recv) {
    const int W = c->dist_world, me = c->dist_rank;
    const long long seq = c->dist_seq++;
    recv.assign((size_t)W, {});
    auto name = [&](int s2, int q) { return c->dist_dir + "/x" + std::to_string(seq) + "." + std::to_string(s2) + "." + std::to_string(q); };
    for (int q = 0; q < W; q++) {
        if (q == me) { recv[me].assign((const char*)send[q].first, (const char*)send[q].first + send[q].second); continue; }
        const std::string fn = name(me, q), tmp = fn + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) return fail(c, -7, "local transport: cannot write " + tmp);
        if (send[q].second && std::fwrite(send[q].first, 1, (size_t)send[q].second, f) != (size_t)send[q].second) { std::fclose(f); return fail(c, -7, "local transport: short write"); }
        std::fclose(f);
        if (std::rename(tmp.c_str(), fn.c_str()) != 0) return fail(c, -7, "local transport: rename failed");
    }
    for (int s2 = 0; s2 < W; s2++) {
        if (s2 == me) continue;
        const std::string fn = name(s2, me);
        FILE* f = nullptr;
        const double t_end = mono_now() + dist_timeout_s();
        while (!(f = std::fopen(fn.c_str(), "rb"))) {
            if (mono_now() > t_end) { c->dist_broken = true; return fail(c, -7, "local transport: timed out waiting for rank " + std::to_string(s2) + " (MIRP_DIST_TIMEOUT_S)"); }
            struct timespec ts = {0, 1000000};
            nanosleep(&ts, nullptr);
        }
        std::fseek(f, 0, SEEK_END);
        const long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        recv[s2].resize((size_t)sz);
        if (sz && std::fread(recv[s2].data(), 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); return fail(c, -7, "local transport: short read"); }
        std::fclose(f);
        std::remove(fn.c_str());
    }
    return 0;
}
}  // namespace

// Completion of what was enqueued on the context's stream, with the deadline: polls the stream and the communicator's asynchronous error state
// instead of blocking in hipStreamSynchronize.  On expiry (or an asynchronous RCCL error) the communicator is ABORTED -- its kernels on the stream
// are torn down, later collectives on this context fail at once (dist_broken) -- and the call returns -7, so the run ends non-zero on every rank
// that is still alive.
static int dist_wait(mirp_ctx* c, const char* what) {
    if (!c->comm || c->dist_world == 1) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    const double t_end = mono_now() + dist_timeout_s();
    long spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return fail(c, -2, std::string(what) + ": " + hipGetErrorString(q));
        ncclResult_t ae = ncclSuccess;
        const ncclResult_t r = g_rccl.CommGetAsyncError((ncclComm_t)c->comm, &ae);
        const bool late = mono_now() > t_end;
        if (r != ncclSuccess || (ae != ncclSuccess && ae != ncclInProgress) || late) {
            const std::string why = late ? std::string("no completion within MIRP_DIST_TIMEOUT_S = ") + std::to_string((int)dist_timeout_s()) + " s (a rank is missing or stuck)"
                                         : std::string("RCCL reported ") + g_rccl.GetErrorString(r != ncclSuccess ? r : ae);
            (void)g_rccl.CommAbort((ncclComm_t)c->comm);
            c->comm = nullptr; c->dist_broken = true;
            return fail(c, -7, std::string(what) + ": " + why + "; communicator aborted");
        }
        if (++spins < 2000) { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }        // 40 ms of fine polling, then 1 ms steps
        else { struct timespec ts = {0, 1000000}; nanosleep(&ts, nullptr); }
    }
}
#define DIST_LIVE(c, what)                                                                                                    \
    do {                                                                                                                      \
        if ((c)->dist_broken) return fail((c), -7, std::string(what) + ": the communicator was aborted by an earlier failure"); \
    } while (0)

#define NCCLCHK(c, call)                                                                                                      \
    do {                                                                                                                      \
        ncclResult_t r_ = (call);                                                                                             \
        if (r_ != ncclSuccess) return fail((c), -7, std::string(#call) + ": " + g_rccl.GetErrorString(r_));                  \
    } while (0)

// Inside an ncclGroupStart / ncclGroupEnd section an error must not return before the group is closed: an open group would silently queue every
// later collective of this thread.  GROUPOP records the first failure and skips the remaining operations; group_end() closes the group and reports.
#define GROUPOP(call)                                                                                                         \
    do {                                                                                                                      \
        if (g_first == ncclSuccess) { g_first = (call); if (g_first != ncclSuccess) g_what = #call; }                         \
    } while (0)
static int group_end(mirp_ctx* c, ncclResult_t g_first, const char* g_what) {
    const ncclResult_t e = g_rccl.GroupEnd();
    if (g_first != ncclSuccess) return fail(c, -7, std::string(g_what) + ": " + g_rccl.GetErrorString(g_first));
    if (e != ncclSuccess) return fail(c, -7, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(e));
    return 0;
}

extern "C" int mirp_dist_unique_id(uint8_t* id) {
    if (!id) return -1;
    static_assert(MIRP_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!g_rccl.load()) return -7;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return -7;
    std::memcpy(id, u.internal, MIRP_DIST_ID_BYTES);
    return 0;
}

extern "C" int mirp_dist_init(mirp_ctx* c, const uint8_t* id, int32_t rank, int32_t world) {
    if (!c) return -1;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(c, -1, "mirp_dist_init: bad argument");
    if (c->comm) return fail(c, -1, "mirp_dist_init: the context already has a communicator");
    if (!g_rccl.load()) return fail(c, -7, g_rccl.err);
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, MIRP_DIST_ID_BYTES);
    ncclComm_t comm = nullptr;
    NCCLCHK(c, g_rccl.CommInitRank(&comm, world, u, rank));
    c->comm = comm; c->dist_rank = rank; c->dist_world = world;
    return 0;
}

extern "C" int mirp_dist_init_local(mirp_ctx* c, const char* dir, int32_t rank, int32_t world) {
    if (!c) return -1;
    if (!dir || !*dir || world < 1 || rank < 0 || rank >= world) return fail(c, -1, "mirp_dist_init_local: bad argument");
    if (c->comm || c->dist_world > 1) return fail(c, -1, "mirp_dist_init_local: the context already has a communicator");
    c->dist_dir = dir; c->dist_rank = rank; c->dist_world = world; c->dist_seq = 0;
    return 0;
}

extern "C" int mirp_dist_finalize(mirp_ctx* c) {
    if (!c) return -1;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    c->dist_rank = 0; c->dist_world = 1; c->dist_dir.clear(); c->dist_broken = false;
    return 0;
}

// What RCCL itself says about the communicator: info[0] = ncclCommCount (ranks it connected), info[1] = ncclCommUserRank, info[2] = ncclCommCuDevice
// (the HIP device it runs on); all -1 without a RCCL communicator (one rank, or the local transport).  bench.py prints them per rank, so that a
// scaling record shows that RCCL -- not a fallback -- saw N ranks on N devices.
extern "C" int mirp_dist_comm_info(mirp_ctx* c, int32_t info[3]) {
    if (!c || !info) return -1;
    info[0] = info[1] = info[2] = -1;
    if (!c->comm) return 0;
    int v = -1;
    NCCLCHK(c, g_rccl.CommCount((ncclComm_t)c->comm, &v)); info[0] = v;
    NCCLCHK(c, g_rccl.CommUserRank((ncclComm_t)c->comm, &v)); info[1] = v;
    NCCLCHK(c, g_rccl.CommCuDevice((ncclComm_t)c->comm, &v)); info[2] = v;
    return 0;
}

extern "C" int mirp_dist_rank(const mirp_ctx* c) { return c ? c->dist_rank : 0; }
extern "C" int mirp_dist_world(const mirp_ctx* c) { return c ? c->dist_world : 1; }

// all ranks: sum of a small int64 vector (in place, host memory); doubles as the barrier
extern "C" int mirp_dist_allreduce_sum(mirp_ctx* c, int64_t* v, int32_t n) {
    if (!c) return -1;
    if (n < 0 || n > 1024 || (n > 0 && !v)) return fail(c, -1, "mirp_dist_allreduce_sum: bad argument");
    if (c->dist_world == 1 || n == 0) return 0;
    DIST_LIVE(c, "mirp_dist_allreduce_sum");
    if (!c->comm) {
        std::vector<long long> all;
        if (int rc = mirp::dist_allgather_ll(c, (const long long*)v, n, all)) return rc;
        for (int k = 0; k < n; k++) { long long t = 0; for (int r = 0; r < c->dist_world; r++) t += all[(size_t)r * n + k]; v[k] = t; }
        return 0;
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (c->dist_tmp.ensure(8 * 1024)) return fail(c, -6, "device allocation failed (dist)");
    HIPCHK(c, hipMemcpyAsync(c->dist_tmp.p, v, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, g_rccl.AllReduce(c->dist_tmp.p, c->dist_tmp.p, (size_t)n, ncclInt64, ncclSum, (ncclComm_t)c->comm, c->stream));
    if (int rc = dist_wait(c, "mirp_dist_allreduce_sum")) return rc;      // before the copy back: a pageable D2H copy blocks in the runtime, without a deadline
    HIPCHK(c, hipMemcpyAsync(v, c->dist_tmp.p, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int mirp_dist_barrier(mirp_ctx* c) {
    int64_t one = 1;
    return mirp_dist_allreduce_sum(c, &one, 1);
}

namespace mirp {
// Every rank learns whether any rank failed in its rank-local preparation (allocation, upload) BEFORE a grouped send / recv: a rank that returned
// early would leave its peers waiting for a block that never comes.  Returns local_rc when this rank failed, -7 naming the first failed rank when
// another one did, 0 when all are ready.
int dist_agree(mirp_ctx* c, int local_rc, const char* what) {
    if (c->dist_world == 1) return local_rc;
    const long long flag = local_rc ? 1 : 0;
    const std::string keep = c->err;
    std::vector<long long> flags;
    if (int rc = dist_allgather_ll(c, &flag, 1, flags)) return local_rc ? (c->err = keep, local_rc) : rc;
    if (local_rc) { c->err = keep; return local_rc; }
    for (int r = 0; r < c->dist_world; r++)
        if (flags[(size_t)r]) return fail(c, -7, std::string(what) + ": rank " + std::to_string(r) + " failed before the exchange");
    return 0;
}

// counts[world] (records of `rec_bytes` bytes each) of every rank, on every rank
int dist_all_counts(mirp_ctx* c, long long mine, std::vector<long long>& counts) { return dist_allgather_ll(c, &mine, 1, counts); }

// Gather of device-resident byte blocks of different sizes to rank dst (rank order): every rank sends `mine` bytes from d_src; on dst, d_dst
// receives sum(counts) bytes.  Grouped ncclSend / ncclRecv on the context's stream (the caller synchronises).
int dist_gatherv_bytes(mirp_ctx* c, const void* d_src, long long mine, int dst, void* d_dst, const std::vector<long long>& counts) {
    if (c->dist_world == 1) {
        if (mine) HIPCHK(c, hipMemcpyAsync(d_dst, d_src, (size_t)mine, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    DIST_LIVE(c, "gather of device blocks");
    if (!c->comm) {      // local transport, host-staged
        std::vector<char> h((size_t)std::max<long long>(mine, 1));
        if (mine) HIPCHK(c, hipMemcpyAsync(h.data(), d_src, (size_t)mine, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::vector<std::pair<const void*, long long>> send((size_t)c->dist_world, {h.data(), 0});
        send[dst].second = mine;
        std::vector<std::vector<char>> recv;
        if (int rc = local_exchange(c, send, recv)) return rc;
        if (c->dist_rank == dst) {
            long long off = 0;
            for (int r = 0; r < c->dist_world; r++) {
                if ((long long)recv[r].size() != counts[r]) return fail(c, -7, "local transport: block size mismatch");
                if (counts[r]) HIPCHK(c, hipMemcpy((char*)d_dst + off, recv[r].data(), (size_t)counts[r], hipMemcpyHostToDevice));
                off += counts[r];
            }
        }
        return 0;
    }
    ncclComm_t comm = (ncclComm_t)c->comm;
    NCCLCHK(c, g_rccl.GroupStart());
    ncclResult_t g_first = ncclSuccess; const char* g_what = "";
    if (c->dist_rank == dst) {
        long long off = 0;
        for (int r = 0; r < c->dist_world; r++) {
            if (r != dst && counts[r]) GROUPOP(g_rccl.Recv((char*)d_dst + off, (size_t)counts[r], ncclUint8, r, comm, c->stream));
            off += counts[r];
        }
    } else if (mine) {
        GROUPOP(g_rccl.Send(d_src, (size_t)mine, ncclUint8, dst, comm, c->stream));
    }
    if (int rc = group_end(c, g_first, g_what)) return rc;
    if (int rc = dist_wait(c, "gather of device blocks")) return rc;
    if (c->dist_rank == dst && mine) {
        long long off = 0;
        for (int r = 0; r < dst; r++) off += counts[r];
        HIPCHK(c, hipMemcpyAsync((char*)d_dst + off, d_src, (size_t)mine, hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}

// All-to-all of device-resident byte blocks: send_off / send_cnt[world] into d_send (bytes), received blocks land in d_recv in source-rank
// order; recv_cnt is filled.  Used by the sharded SAM ingest to route records to the rank that owns their contig.
int dist_alltoallv_bytes(mirp_ctx* c, const void* d_send, const std::vector<long long>& send_off, const std::vector<long long>& send_cnt, void* d_recv,
                         const std::vector<long long>& recv_off, const std::vector<long long>& recv_cnt) {
    const int W = c->dist_world, me = c->dist_rank;
    if (W == 1) {
        if (send_cnt[0]) HIPCHK(c, hipMemcpyAsync((char*)d_recv + recv_off[0], (const char*)d_send + send_off[0], (size_t)send_cnt[0], hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    DIST_LIVE(c, "all-to-all of device blocks");
    if (!c->comm) {      // local transport, host-staged
        long long tot = 0;
        for (int r = 0; r < W; r++) tot = std::max(tot, send_off[r] + send_cnt[r]);
        std::vector<char> h((size_t)std::max<long long>(tot, 1));
        if (tot) HIPCHK(c, hipMemcpyAsync(h.data(), d_send, (size_t)tot, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::vector<std::pair<const void*, long long>> send((size_t)W);
        for (int r = 0; r < W; r++) send[r] = {h.data() + send_off[r], send_cnt[r]};
        std::vector<std::vector<char>> recv;
        if (int rc = local_exchange(c, send, recv)) return rc;
        for (int r = 0; r < W; r++) {
            if ((long long)recv[r].size() != recv_cnt[r]) return fail(c, -7, "local transport: block size mismatch");
            if (recv_cnt[r]) HIPCHK(c, hipMemcpy((char*)d_recv + recv_off[r], recv[r].data(), (size_t)recv_cnt[r], hipMemcpyHostToDevice));
        }
        return 0;
    }
    ncclComm_t comm = (ncclComm_t)c->comm;
    NCCLCHK(c, g_rccl.GroupStart());
    ncclResult_t g_first = ncclSuccess; const char* g_what = "";
    for (int r = 0; r < W; r++) {
        if (r == me) continue;
        if (send_cnt[r]) GROUPOP(g_rccl.Send((const char*)d_send + send_off[r], (size_t)send_cnt[r], ncclUint8, r, comm, c->stream));
        if (recv_cnt[r]) GROUPOP(g_rccl.Recv((char*)d_recv + recv_off[r], (size_t)recv_cnt[r], ncclUint8, r, comm, c->stream));
    }
    if (int rc = group_end(c, g_first, g_what)) return rc;
    if (int rc = dist_wait(c, "all-to-all of device blocks")) return rc;
    if (send_cnt[me]) HIPCHK(c, hipMemcpyAsync((char*)d_recv + recv_off[me], (const char*)d_send + send_off[me], (size_t)send_cnt[me], hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

// every rank's `mine[n]` int64 vector on every rank: out[r * n + k]
int dist_allgather_ll(mirp_ctx* c, const long long* mine, int n, std::vector<long long>& out) {
    const int W = c->dist_world;
    out.assign((size_t)W * n, 0);
    if (W == 1) { for (int k = 0; k < n; k++) out[k] = mine[k]; return 0; }
    DIST_LIVE(c, "all-gather of counts");
    if (!c->comm) {
        std::vector<std::pair<const void*, long long>> send((size_t)W, {mine, 8LL * n});
        std::vector<std::vector<char>> recv;
        if (int rc = local_exchange(c, send, recv)) return rc;
        for (int r = 0; r < W; r++) {
            if ((long long)recv[r].size() != 8LL * n) return fail(c, -7, "local transport: block size mismatch");
            std::memcpy(out.data() + (size_t)r * n, recv[r].data(), 8 * (size_t)n);
        }
        return 0;
    }
    if (c->dist_tmp.ensure(8 * (size_t)(W + 1) * n + 64)) return fail(c, -6, "device allocation failed (dist)");
    long long* d = (long long*)c->dist_tmp.p;
    HIPCHK(c, hipMemcpyAsync(d + (size_t)W * n, mine, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, g_rccl.AllGather(d + (size_t)W * n, d, (size_t)n, ncclInt64, (ncclComm_t)c->comm, c->stream));
    if (int rc = dist_wait(c, "all-gather of counts")) return rc;
    HIPCHK(c, hipMemcpyAsync(out.data(), d, 8 * (size_t)W * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
}  // namespace mirp

// Generic gather of fixed-size host records to rank dst, in rank order (host buffers in and out; staged through the device for RCCL).
extern "C" int mirp_gather_records(mirp_ctx* c, const void* rec, int64_t n, int32_t rec_bytes, int32_t dst, void** out, int64_t* n_out) {
    if (!c) return -1;
    if (n < 0 || rec_bytes < 1 || (n > 0 && !rec) || !out || !n_out || dst < 0 || dst >= c->dist_world) return fail(c, -1, "mirp_gather_records: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    *out = nullptr; *n_out = 0;
    std::vector<long long> counts;
    if (int rc = mirp::dist_all_counts(c, (long long)n * rec_bytes, counts)) return rc;
    long long total = 0;
    for (long long x : counts) total += x;
    TmpDevice T;
    void* d_src = T.get((size_t)n * rec_bytes + 16);
    void* d_dst = c->dist_rank == dst ? T.get((size_t)total + 16) : nullptr;
    void* h = c->dist_rank == dst ? std::malloc((size_t)std::max<long long>(total, 1)) : nullptr;
    int prep_rc = 0;
    if (!d_src || (c->dist_rank == dst && (!d_dst || !h))) prep_rc = fail(c, -6, "allocation failed (gather)");
    if (!prep_rc && n) {
        hipError_t he = hipMemcpyAsync(d_src, rec, (size_t)n * rec_bytes, hipMemcpyHostToDevice, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) prep_rc = fail(c, -2, std::string("mirp_gather_records: upload failed: ") + hipGetErrorString(he));
    }
    if (int rc = mirp::dist_agree(c, prep_rc, "mirp_gather_records")) { std::free(h); return rc; }
    if (int rc = mirp::dist_gatherv_bytes(c, d_src, (long long)n * rec_bytes, dst, d_dst, counts)) { std::free(h); return rc; }
    if (c->dist_rank == dst) {
        hipError_t he = total ? hipMemcpyAsync(h, d_dst, (size_t)total, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { std::free(h); return fail(c, -2, std::string("mirp_gather_records: ") + hipGetErrorString(he)); }
        *out = h; *n_out = total / rec_bytes;
    } else {
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

extern "C" int mirp_exchange_bytes(mirp_ctx* c, const void* send, const int64_t* send_cnt, void** recv, int64_t* recv_cnt) {
    if (!c) return -1;
    if (!send_cnt || !recv || !recv_cnt) return fail(c, -1, "mirp_exchange_bytes: null argument");
    const int W = c->dist_world;
    HIPCHK(c, hipSetDevice(c->device));
    *recv = nullptr;
    std::vector<long long> scnt((size_t)W), soff((size_t)W), all;
    long long stot = 0;
    for (int r = 0; r < W; r++) {
        if (send_cnt[r] < 0 || (send_cnt[r] > 0 && !send)) return fail(c, -1, "mirp_exchange_bytes: bad argument");
        scnt[r] = send_cnt[r]; soff[r] = stot; stot += (send_cnt[r] + 15) & ~15LL;
    }
    if (int rc = mirp::dist_allgather_ll(c, scnt.data(), W, all)) return rc;      // all[s * W + q] = bytes rank s sends to rank q
    std::vector<long long> rcnt((size_t)W), roff((size_t)W);
    long long rtot = 0, rsum = 0;
    for (int s2 = 0; s2 < W; s2++) { rcnt[s2] = all[(size_t)s2 * W + c->dist_rank]; roff[s2] = rtot; rtot += (rcnt[s2] + 15) & ~15LL; rsum += rcnt[s2]; }
    // allocations and the upload can fail on one rank only: agree before the grouped send / recv (as the sharded ingest does)
    TmpDevice T;
    char* d_send = (char*)T.get((size_t)stot + 16);
    char* d_recv = (char*)T.get((size_t)rtot + 16);
    char* h = (char*)std::malloc((size_t)std::max<long long>(rsum, 1));
    int prep_rc = 0;
    if (!d_send || !d_recv || !h) prep_rc = fail(c, -6, "allocation failed (exchange)");
    if (!prep_rc) {
        hipError_t he = hipSuccess;
        for (int r = 0; r < W && he == hipSuccess; r++) {
            const long long src_off = [&] { long long o = 0; for (int x = 0; x < r; x++) o += send_cnt[x]; return o; }();
            if (scnt[r]) he = hipMemcpyAsync(d_send + soff[r], (const char*)send + src_off, (size_t)scnt[r], hipMemcpyHostToDevice, c->stream);
        }
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) prep_rc = fail(c, -2, std::string("mirp_exchange_bytes: upload failed: ") + hipGetErrorString(he));
    }
    if (int rc = mirp::dist_agree(c, prep_rc, "mirp_exchange_bytes")) { std::free(h); return rc; }
    if (int rc = mirp::dist_alltoallv_bytes(c, d_send, soff, scnt, d_recv, roff, rcnt)) { std::free(h); return rc; }
    hipError_t he = hipSuccess;
    long long o = 0;
    for (int s2 = 0; s2 < W && he == hipSuccess; s2++) {
        if (rcnt[s2]) he = hipMemcpyAsync(h + o, d_recv + roff[s2], (size_t)rcnt[s2], hipMemcpyDeviceToHost, c->stream);
        recv_cnt[s2] = rcnt[s2]; o += rcnt[s2];
    }
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he != hipSuccess) { std::free(h); return fail(c, -2, std::string("mirp_exchange_bytes: download failed: ") + hipGetErrorString(he)); }
    *recv = h;
    return 0;
}

// The exchange step of the path: the loci list of the last mirp_predict of every rank, gathered to rank dst in rank order straight
// out of the device-resident result arrays (64-byte MirpMirna records + structure text rows).  Ranks other than dst get n_result = 0.
extern "C" int mirp_gather_loci(mirp_ctx* c, int32_t dst, MirpMirna** result, int64_t* n_result, char** ss_text, int32_t* ss_stride) {
    if (!c) return -1;
    if (!result || !n_result || !ss_text || !ss_stride || dst < 0 || dst >= c->dist_world) return fail(c, -1, "mirp_gather_loci: bad argument");
    if (!c->have_result) return fail(c, -1, "mirp_gather_loci: run mirp_predict first");
    HIPCHK(c, hipSetDevice(c->device));
    *result = nullptr; *n_result = 0; *ss_text = nullptr; *ss_stride = c->fold_stride;
    const long long mine = c->n_result;
    long long v[2] = {mine, c->fold_stride};
    std::vector<long long> all;
    if (int rc = mirp::dist_allgather_ll(c, v, 2, all)) return rc;
    std::vector<long long> crec((size_t)c->dist_world), ctxt((size_t)c->dist_world);
    long long total = 0;
    for (int r = 0; r < c->dist_world; r++) {
        if (all[2 * r + 1] != c->fold_stride) return fail(c, -1, "mirp_gather_loci: the ranks folded with different PRECURSOR_LEN (structure text stride differs)");
        crec[r] = all[2 * r] * (long long)sizeof(MirpMirna); ctxt[r] = all[2 * r] * c->fold_stride; total += all[2 * r];
    }
    TmpDevice T;
    const bool root = c->dist_rank == dst;
    void* d_rec = root ? T.get((size_t)total * sizeof(MirpMirna) + 16) : nullptr;
    void* d_txt = root ? T.get((size_t)total * c->fold_stride + 16) : nullptr;
    MirpMirna* hr = root ? (MirpMirna*)std::calloc((size_t)std::max<long long>(total, 1), sizeof(MirpMirna)) : nullptr;
    char* ht = root ? (char*)std::calloc((size_t)std::max<long long>(total, 1), (size_t)c->fold_stride) : nullptr;
    // a root that cannot allocate must not leave the other ranks in ncclSend: agree first
    const int prep_rc = (root && (!d_rec || !d_txt || !hr || !ht)) ? fail(c, -6, "allocation failed (gather of the loci list)") : 0;
    if (int rc = mirp::dist_agree(c, prep_rc, "mirp_gather_loci")) { std::free(hr); std::free(ht); return rc; }
    if (int rc = mirp::dist_gatherv_bytes(c, c->p_res.p, mine * (long long)sizeof(MirpMirna), dst, d_rec, crec)) { std::free(hr); std::free(ht); return rc; }
    if (int rc = mirp::dist_gatherv_bytes(c, c->p_text.p, mine * c->fold_stride, dst, d_txt, ctxt)) { std::free(hr); std::free(ht); return rc; }
    if (root) {
        hipError_t he = hipSuccess;
        if (total) {
            he = hipMemcpyAsync(hr, d_rec, (size_t)total * sizeof(MirpMirna), hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(ht, d_txt, (size_t)total * c->fold_stride, hipMemcpyDeviceToHost, c->stream);
        }
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { std::free(hr); std::free(ht); return fail(c, -2, std::string("mirp_gather_loci: ") + hipGetErrorString(he)); }
        *result = hr; *n_result = total; *ss_text = ht;
    } else {
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}
