// TEST INFRASTRUCTURE, never shipped, never loaded by the product outside tests/test_gpu_parity.py::test_comm_ranks_exchange_through_a_
// loopback_transport: a stand-in for librccl.so.1 that lets SEVERAL ranks of rto_comm live in ONE process on ONE GPU (RCCL itself
// refuses two ranks on one device: "Duplicate GPU detected"), so that the product's own rto_comm_submit / comm_exchange / pack /
// assemble code runs at world sizes 2..8 with real bytes going from rank r's buffers into rank 0's -- everything except RCCL's
// transport.  The test puts the directory of the built librccl.so.1 in front of LD_LIBRARY_PATH of a child process that never
// imports torch (torch maps the real library under the same SONAME); rto_comm.inc opens "librccl.so.1" with dlopen and finds this.
//
// What it implements: point-to-point ncclSend / ncclRecv between the communicators of one "clique" (one ncclUniqueId), matched in
// posting order per (source, destination) pair like NCCL's, element counts checked to agree, the copy a hipMemcpyAsync on the
// RECEIVER's stream behind an event the sender's stream recorded when the send was posted; the sender's stream then waits for the
// copy.  Two modes:
//  * default (one host thread drives all ranks: ranks N-1..1 submit first, rank 0 last, every rank flushes before the next batch): a
//    send whose receive has not been posted yet waits in a list, no call blocks;
//  * RTO_RCCL_SHIM_RENDEZVOUS=1 (one host thread PER rank, batches pipelined without flushes): ncclGroupEnd returns only when each
//    of its operations has met its counterpart (30 s at most: then ncclSystemError, never a hang) -- the copy is then queued on the
//    receiver's stream in front of whatever the receiver's thread queues next (its assembly), and the sender's stream waits for it in
//    front of whatever the sender queues next (the event that frees its buffers): the ordering NCCL's kernels give.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

struct ncclComm {
    int rank = 0, world = 1;
    uint64_t clique = 0;
};

namespace {

struct Op {
    bool send;
    const void* sbuf;
    void* rbuf;
    size_t count;
    int peer;
    ncclComm* comm;
    hipStream_t stream;
    hipEvent_t posted;      // recorded on `stream` when the operation was posted
    int* met = nullptr;     // rendezvous mode: set to 1 (2: failed) by the thread that completes the pair
};

std::mutex g_mu;
std::condition_variable g_cv;
bool rendezvous() { static const bool on = [] { const char* e = std::getenv("RTO_RCCL_SHIM_RENDEZVOUS"); return e && *e == '1'; }(); return on; }
std::deque<Op> g_waiting;           // posted, counterpart not yet seen
thread_local int t_depth = 0;
thread_local std::vector<Op> t_group;
uint64_t g_nextClique = 1;
int g_sizeMismatch = 0;

uint64_t clique_of(const ncclUniqueId& id) {
    uint64_t v;
    std::memcpy(&v, id.internal, sizeof v);
    return v;
}

ncclResult_t copy_now(const Op& s, const Op& r) {
    if (s.count != r.count) { g_sizeMismatch++; return ncclInvalidArgument; }
    if (hipStreamWaitEvent(r.stream, s.posted, 0) != hipSuccess) return ncclUnhandledCudaError;
    if (s.count && hipMemcpyAsync(r.rbuf, s.sbuf, s.count * sizeof(float), hipMemcpyDeviceToDevice, r.stream) != hipSuccess) return ncclUnhandledCudaError;
    hipEvent_t done;
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(done, r.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamWaitEvent(s.stream, done, 0) != hipSuccess) return ncclUnhandledCudaError;     // the sender's stream goes on behind the copy
    (void)hipEventDestroy(done);            // released when it has fired
    (void)hipEventDestroy(s.posted);
    (void)hipEventDestroy(r.posted);
    return ncclSuccess;
}

bool matches(const Op& a, const Op& b) {    // a: send, b: recv
    return a.comm->clique == b.comm->clique && a.peer == b.comm->rank && b.peer == a.comm->rank;
}

ncclResult_t post(Op op) {
    if (hipEventCreateWithFlags(&op.posted, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(op.posted, op.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lock(g_mu);
    for (auto it = g_waiting.begin(); it != g_waiting.end(); ++it) {
        if (it->send == op.send) continue;
        const Op& s = op.send ? op : *it;
        const Op& r = op.send ? *it : op;
        if (!matches(s, r)) continue;
        const ncclResult_t rc = copy_now(s, r);
        if (it->met) *it->met = rc == ncclSuccess ? 1 : 2;
        g_waiting.erase(it);
        g_cv.notify_all();
        return rc;
    }
    if (!rendezvous()) { g_waiting.push_back(op); return ncclSuccess; }
    int met = 0;
    op.met = &met;
    g_waiting.push_back(op);
    const bool ok = g_cv.wait_for(lock, std::chrono::seconds(30), [&] { return met != 0; });
    if (!ok) {      // the counterpart never came: take the operation back, fail loudly
        for (auto it = g_waiting.begin(); it != g_waiting.end(); ++it)
            if (it->met == &met) { g_waiting.erase(it); break; }
        return ncclSystemError;
    }
    return met == 1 ? ncclSuccess : ncclInvalidArgument;
}

ncclResult_t enqueue(const Op& op) {
    if (t_depth > 0) { t_group.push_back(op); return ncclSuccess; }
    return post(op);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    std::lock_guard<std::mutex> lock(g_mu);
    const uint64_t v = 0x5348494d00000000ull | g_nextClique++;      // "SHIM"
    std::memcpy(id->internal, &v, sizeof v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->world = nranks; c->clique = clique_of(id);
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    (void)devlist;
    if (!comms || ndev < 1) return ncclInvalidArgument;
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    for (int i = 0; i < ndev; i++) ncclCommInitRank(&comms[i], ndev, id, i);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t* asyncError) { if (asyncError) *asyncError = ncclSuccess; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { if (!comm || !count) return ncclInvalidArgument; *count = comm->world; return ncclSuccess; }

ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_group);
    ncclResult_t rc = ncclSuccess;
    // a rank's send to itself meets its own receive inside the group (in rendezvous mode it would otherwise wait for itself)
    std::vector<char> done(ops.size(), 0);
    for (size_t i = 0; i < ops.size(); i++) {
        if (done[i] || !ops[i].send || ops[i].peer != ops[i].comm->rank) continue;
        for (size_t j = 0; j < ops.size(); j++) {
            if (done[j] || ops[j].send || ops[j].comm != ops[i].comm || ops[j].peer != ops[i].comm->rank) continue;
            Op s = ops[i], r = ops[j];
            if (hipEventCreateWithFlags(&s.posted, hipEventDisableTiming) != hipSuccess || hipEventRecord(s.posted, s.stream) != hipSuccess ||
                hipEventCreateWithFlags(&r.posted, hipEventDisableTiming) != hipSuccess || hipEventRecord(r.posted, r.stream) != hipSuccess) return ncclUnhandledCudaError;
            const ncclResult_t q = copy_now(s, r);
            if (q != ncclSuccess) rc = q;
            done[i] = done[j] = 1;
            break;
        }
    }
    for (size_t i = 0; i < ops.size(); i++) {
        if (done[i]) continue;
        const ncclResult_t r = post(ops[i]);
        if (r != ncclSuccess) rc = r;
    }
    return rc;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || datatype != ncclFloat32 || peer < 0 || peer >= comm->world) return ncclInvalidArgument;
    return enqueue(Op{ true, sendbuff, nullptr, count, peer, comm, stream, nullptr });
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || datatype != ncclFloat32 || peer < 0 || peer >= comm->world) return ncclInvalidArgument;
    return enqueue(Op{ false, nullptr, recvbuff, count, peer, comm, stream, nullptr });
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "shim: success";
        case ncclInvalidArgument: return "shim: invalid argument (or: the two sides of a send / receive disagree on the element count)";
        case ncclInvalidUsage: return "shim: invalid usage";
        case ncclUnhandledCudaError: return "shim: HIP error";
        case ncclSystemError: return "shim: the counterpart of a send / receive never came (30 s)";
        default: return "shim: error";
    }
}

// for the test: operations still waiting for their counterpart (must be 0 after every batch), element-count disagreements seen
int rccl_shim_pending() { std::lock_guard<std::mutex> lock(g_mu); return (int)g_waiting.size(); }
int rccl_shim_size_mismatches() { return g_sizeMismatch; }

}  // extern "C"
