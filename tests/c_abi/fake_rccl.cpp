// fake_rccl.cpp — TEST INFRASTRUCTURE, not product.  A single-process stand-in for the eight RCCL entry points
// bvh_amd/csrc/comm.hip uses, loaded through BVHGPU_RCCL_LIB, so that the multi-GPU PROTOCOL of the C ABI (who sends
// what when, status header, peers' trees, error and rebroadcast paths) runs with several "ranks" on the ONE GPU a test box
// has — real RCCL refuses two ranks on one device, and this container can reach no multi-GPU node.  It does not test
// RCCL itself: a broadcast here is a device-to-device copy ordered with events (root stream → peer stream → root stream).
//
//   ncclCommInitAll(comms, ndev, devs)   ndev ranks of one world driven by ONE thread, any devices (also all the same one)
//   ncclCommInitRank(nranks, id, rank)   one rank per THREAD of this process (the process-per-GPU form of the C ABI, with threads
//                                        standing in for processes): the ranks that present the same unique id form a world; a
//                                        group's collectives run when every rank of the world has closed its group (a barrier)
//   ncclGroupStart / ncclGroupEnd        the k-th broadcast each rank enqueued in the group forms the k-th collective
//   ncclBroadcast                        inside a group: recorded; outside: a group of one call
// Mismatched collectives (different counts / roots / number of calls across the ranks of a group) return
// ncclInvalidUsage — what would be a hang or corruption with the real library is a test failure here.
// build: hipcc -shared -fPIC -o libfakerccl.so fake_rccl.cpp   (tests/conftest.py does it)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Op;
struct World {
    int nranks; int live;
    bool threaded = false;                      // ranks are threads (ncclCommInitRank): groups meet at a barrier
    std::mutex mu; std::condition_variable cv;
    std::vector<std::vector<Op>> pending;       // per rank: the ops of the group it has closed
    int arrived = 0; unsigned long long round = 0; ncclResult_t result = ncclSuccess;
};
struct Comm { World* world; int rank; int device; };
struct Op { Comm* comm; const void* send; void* recv; size_t bytes; int root; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
unsigned long long g_calls = 0;   // collectives completed (the tests read it through fake_rccl_collectives)

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

ncclResult_t run_ops(std::vector<Op>& ops) {
    // per world: the ranks' op lists in call order
    std::map<World*, std::map<int, std::vector<Op*>>> by_world;
    for (auto& o : ops) by_world[o.comm->world][o.comm->rank].push_back(&o);
    for (auto& kv : by_world) {
        World* w = kv.first;
        auto& ranks = kv.second;
        if ((int)ranks.size() != w->nranks) { fprintf(stderr, "fake_rccl: %zu of %d ranks took part in a group\n", ranks.size(), w->nranks); return ncclInvalidUsage; }
        const size_t ncoll = ranks.begin()->second.size();
        for (auto& r : ranks) if (r.second.size() != ncoll) { fprintf(stderr, "fake_rccl: ranks issued different numbers of broadcasts\n"); return ncclInvalidUsage; }
        for (size_t k = 0; k < ncoll; k++) {
            const Op* first = ranks.begin()->second[k];
            if (first->root < 0 || first->root >= w->nranks) return ncclInvalidArgument;
            const Op* root = ranks[first->root][k];
            for (auto& r : ranks) {
                const Op* o = r.second[k];
                if (o->bytes != first->bytes || o->root != first->root) { fprintf(stderr, "fake_rccl: broadcast %zu differs between ranks (bytes %zu / %zu, root %d / %d)\n", k, o->bytes, first->bytes, o->root, first->root); return ncclInvalidUsage; }
            }
            hipEvent_t ready;
            if (hipSetDevice(root->comm->device) != hipSuccess) return ncclUnhandledCudaError;
            if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
            if (hipEventRecord(ready, root->stream) != hipSuccess) return ncclUnhandledCudaError;
            if (root->recv != root->send && root->bytes)
                if (hipMemcpyAsync(root->recv, root->send, root->bytes, hipMemcpyDeviceToDevice, root->stream) != hipSuccess) return ncclUnhandledCudaError;
            for (auto& r : ranks) {
                const Op* o = r.second[k];
                if (o == root) continue;
                if (hipSetDevice(o->comm->device) != hipSuccess) return ncclUnhandledCudaError;
                if (hipStreamWaitEvent(o->stream, ready, 0) != hipSuccess) return ncclUnhandledCudaError;
                if (o->bytes && hipMemcpyAsync(o->recv, root->send, o->bytes, hipMemcpyDefault, o->stream) != hipSuccess) return ncclUnhandledCudaError;
                hipEvent_t done;   // the root's stream may overwrite its send buffer only after the peer has read it
                if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
                if (hipEventRecord(done, o->stream) != hipSuccess) return ncclUnhandledCudaError;
                if (hipStreamWaitEvent(root->stream, done, 0) != hipSuccess) return ncclUnhandledCudaError;
                (void)hipEventDestroy(done);
            }
            (void)hipEventDestroy(ready);
            g_calls++;
        }
    }
    return ncclSuccess;
}
}  // namespace

extern "C" {

static std::mutex g_mu;
static std::map<std::string, World*> g_worlds;   // unique id → world being formed by ncclCommInitRank
static unsigned long long g_ids = 0;

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "fake_rccl-%llu", ++g_ids);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    std::lock_guard<std::mutex> lk(g_mu);
    const std::string key(id.internal, sizeof id.internal);
    World*& w = g_worlds[key];
    if (!w) { w = new World(); w->nranks = nranks; w->live = nranks; w->threaded = nranks > 1; w->pending.resize(nranks); }
    if (w->nranks != nranks) return ncclInvalidUsage;
    *comm = reinterpret_cast<ncclComm_t>(new Comm{w, rank, dev});
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devs) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    World* w = new World();
    w->nranks = ndev; w->live = ndev;
    for (int i = 0; i < ndev; i++) comms[i] = reinterpret_cast<ncclComm_t>(new Comm{w, i, devs ? devs[i] : i});
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    if (--c->world->live == 0) {
        for (auto it = g_worlds.begin(); it != g_worlds.end(); ++it) if (it->second == c->world) { g_worlds.erase(it); break; }
        delete c->world;
    }
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
// a closed group: single-thread worlds run at once; a threaded world's ranks meet here — the last one to arrive runs everybody's ops
static ncclResult_t close_group(std::vector<Op>& ops) {
    std::vector<Op> local;
    std::map<World*, std::vector<Op>> threaded;
    for (auto& o : ops) (o.comm->world->threaded ? threaded[o.comm->world] : local).push_back(o);
    ncclResult_t rc = local.empty() ? ncclSuccess : run_ops(local);
    for (auto& kv : threaded) {
        World* w = kv.first;
        const int rank = kv.second[0].comm->rank;
        std::unique_lock<std::mutex> lk(w->mu);
        w->pending[rank] = kv.second;
        const unsigned long long my_round = w->round;
        if (++w->arrived == w->nranks) {
            std::vector<Op> all;
            for (auto& v : w->pending) { all.insert(all.end(), v.begin(), v.end()); v.clear(); }
            w->result = run_ops(all);
            w->arrived = 0; w->round++;
            w->cv.notify_all();
        } else {
            w->cv.wait(lk, [&] { return w->round != my_round; });
        }
        if (w->result != ncclSuccess) rc = w->result;
    }
    return rc;
}
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return close_group(ops);
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || (count && (!send || !recv))) return ncclInvalidArgument;
    g_ops.push_back(Op{c, send, recv, count * type_size(type), root, stream});
    if (g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return close_group(ops);
}
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl)";
        case ncclInvalidUsage: return "invalid usage (fake_rccl): the ranks' collectives do not match";
        default: return "error (fake_rccl)";
    }
}
unsigned long long fake_rccl_collectives(void) { return g_calls; }

}  // extern "C"
