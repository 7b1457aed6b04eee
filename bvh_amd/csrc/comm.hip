// comm.hip — the one exchange step of the path on several GPUs: the root's flattened tree reaches the
// peers in RCCL broadcasts over xGMI, straight from the root tree's own HBM buffers into the peers' (no staging blob).
// Two ways to form the communicator, as RCCL offers them:
//   bvhgpu_comm_init_all   one process drives ndev GPUs (ncclCommInitAll; one ctx per device)
//   bvhgpu_comm_init_rank  one process per GPU (ncclCommInitRank; the launcher carries the 128-byte id to the peers)
// What travels: a 64-byte status header, the folded traversal array, the shape AABBs, the binary LDS slot table and, if
// asked for, the triangle vertices.  The wide nodes (30 MB at 120 k triangles) are NOT sent: every peer rebuilds them from
// the traversal array with one kernel (flatten.hip k_wide) — cheaper than moving them over a 150 GB/s link.  Rays never
// travel: each GPU generates / owns its shard, and hit lists stay where they were produced.
//
// Protocol rules (round 3; the round-2 version could leave the peers blocked inside ncclBroadcast):
//  * A collective call is never abandoned half-way by one rank.  Whatever the root finds wrong with its own tree (NULL, not
//    flattened, another dtype / shape count than announced, invalid input found by the build) it still issues every
//    broadcast of the call; the status header tells the peers that nothing usable arrived, and they return an error from
//    their wait instead of blocking for ever.
//  * bvhgpu_bcast_known has no host round trip on any rank, also when the root's build is still in flight on its stream
//    (bvhgpu_rebuild_flat_async): the header is then composed ON THE DEVICE from the status word the build's last kernel
//    leaves in HBM.  The peers copy the received header to pinned memory behind the receive and look at it when their
//    tree is first waited for (bvhgpu_tree_wait / bvhgpu_hits_wait / any entry point that inspects the tree).
//  * What the header carries besides the sizes: `exact_only` (a split without SAH winner: the receiver must not walk the
//    tree wide, traverse.hip) and "the optimistic build was not complete" (unbalanced tree on a first build) — then the
//    root's wait and the peers' waits all return BVHGPU_REBROADCAST and every rank repeats the call.
//  * RCCL is loaded on first use (dlopen): a single-GPU consumer of libbvh_mi355x.so does not need librccl at all, and a
//    process that already holds a copy (PyTorch bundles one) shares it.  No librccl → BVHGPU_RCCL_ERROR.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and constants only: every call goes through the table below

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "engine.hpp"

using namespace bvhgpu;

struct bvhgpu_comm {
    std::vector<ncclComm_t> comms;     // one per local device
    std::vector<bvhgpu_ctx*> ctxs;
    std::vector<void*> hdr_dev;        // 64-byte header buffer in each device's HBM
    int nranks = 0;                    // size of the communicator
    int first_rank = 0;                // rank of local device 0 (init_all: 0)
};

namespace {

// ---- RCCL, loaded lazily ----------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    std::string err;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;   // optional: reported by bvhgpu_rccl_info
    std::string path;      // the file the entry points came from (dladdr), for bvhgpu_rccl_info
    bool shared = false;   // true: a copy the process already held (RTLD_NOLOAD), no second RCCL was loaded
    bool ok() const { return lib != nullptr; }
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // BVHGPU_RCCL_LIB: an explicit library (the tests load a single-process stand-in through it); otherwise a copy the
        // process already holds (PyTorch-ROCm bundles librccl.so.1 — two RCCLs in one process would each open the devices),
        // then the loader's search path, then the ROCm installation
        std::vector<std::string> names;
        if (const char* e = std::getenv("BVHGPU_RCCL_LIB")) names.push_back(e);
        void* h = nullptr;
        if (names.empty())   // (RTLD_NOLOAD matches by SONAME or by the name the copy was loaded under: try both spellings)
            for (const char* n : {"librccl.so.1", "librccl.so"}) {
                h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
                if (h) { api.shared = true; break; }
            }
        if (!h) {
            if (names.empty()) {
                names = {"librccl.so.1", "librccl.so"};
                const char* rp = std::getenv("ROCM_PATH");
                names.push_back(std::string(rp ? rp : "/opt/rocm") + "/lib/librccl.so.1");
            }
            for (const auto& n : names) {
                h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
                if (h) break;
                if (const char* d = dlerror()) api.err = d;
            }
        }
        if (!h) { if (api.err.empty()) api.err = "librccl.so.1 not found"; return; }
        bool all = true;
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) { all = false; api.err = std::string("missing symbol ") + n; } return p; };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(h, "ncclGetVersion"));
        if (all) {
            api.lib = h;
            Dl_info di;
            if (dladdr(reinterpret_cast<void*>(api.Broadcast), &di) && di.dli_fname) api.path = di.dli_fname;
            if (std::getenv("BVHGPU_RCCL_DEBUG"))
                fprintf(stderr, "bvhgpu: RCCL entry points from %s (%s)\n", api.path.c_str(), api.shared ? "already loaded in this process" : "loaded by libbvh_mi355x");
        }
    });
    return api;
}

struct RcclFail { ncclResult_t err; const char* what; int line; };
#define BVH_RCCL(x)                                              \
    do {                                                         \
        ncclResult_t _r = (x);                                   \
        if (_r != ncclSuccess) throw RcclFail{_r, #x, __LINE__}; \
    } while (0)

int comm_fail(bvhgpu_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return status;
}

template <typename F> int comm_guarded(bvhgpu_ctx* ctx, F&& f) {
    try {
        return f();
    } catch (const RcclFail& e) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s failed: %s (comm.hip line %d)", e.what, rccl().GetErrorString ? rccl().GetErrorString(e.err) : "?", e.line);
        return comm_fail(ctx, BVHGPU_RCCL_ERROR, buf);
    } catch (const HipFail& e) {
        char buf[512];
        if (e.what && std::strcmp(e.what, "NONFINITE") == 0)
            return comm_fail(ctx, BVHGPU_INVALID_ARG, "shape AABBs contain NaN or infinity (bvh_node.rs:214-217): nothing was built, nothing usable was sent");
        if (e.what && std::strcmp(e.what, "RECV_REBROADCAST") == 0)
            return comm_fail(ctx, BVHGPU_REBROADCAST, "the tree was received from a root whose optimistic build was not complete: every rank calls bvhgpu_bcast_known again");
        if (e.what && std::strncmp(e.what, "RECV_", 5) == 0)
            return comm_fail(ctx, BVHGPU_INVALID_ARG, "the tree was received from a root that had no valid tree to send");
        snprintf(buf, sizeof buf, "%s failed: %s (comm.hip line %d)", e.what, hipGetErrorString(e.err), e.line);
        return comm_fail(ctx, e.err == hipErrorOutOfMemory ? BVHGPU_OOM : BVHGPU_HIP_ERROR, buf);
    } catch (const std::bad_alloc&) {
        return comm_fail(ctx, BVHGPU_OOM, "host allocation failed");
    } catch (...) {
        return comm_fail(ctx, BVHGPU_HIP_ERROR, "unknown exception");
    }
}
int need_rccl(bvhgpu_ctx* ctx) {
    if (rccl().ok()) return BVHGPU_OK;
    return comm_fail(ctx, BVHGPU_RCCL_ERROR, "RCCL is not available: " + rccl().err + " (set BVHGPU_RCCL_LIB or install librccl.so.1; single-GPU use does not need it)");
}

// ---- the status header --------------------------------------------------------------------------
constexpr uint32_t BH_GOOD = 0u, BH_INVALID = 1u, BH_UNFINISHED = 2u;
struct BcastHeader {   // 64 bytes
    uint32_t magic, status;            // status: BH_*
    uint32_t dtype, flags;             // flags bit 0: exact_only
    uint64_t n, n_trav;
    uint32_t unfolded, has_tris, has_slots, _pad;
    uint64_t _r[2];
};
static_assert(sizeof(BcastHeader) == 64, "bcast header");
constexpr uint32_t BCAST_MAGIC = 0x42564843u;   // "BVHC" (round 2: BVHB, without status / flags)

// The header is written into HBM by a kernel, so that it is ordered on the root's stream behind the build it may describe.
// bstat != NULL: the build's outcome is not known to the host yet — it is taken from the word k_flatten's publishing block left.
__global__ void k_bcast_header(BcastHeader* dst, BcastHeader h, const uint32_t* __restrict__ bstat) {
    if (threadIdx.x != 0) return;
    if (bstat && h.status == BH_GOOD) {   // (what the host already found wrong with the root's tree stays: the root does not join a rebroadcast)
        const uint32_t s = bstat[0];
        if (s & BSTAT_NONFINITE) h.status = BH_INVALID;
        else if (s & BSTAT_UNFINISHED) h.status = BH_UNFINISHED;
        if (s & BSTAT_EMPTY_SPLIT) h.flags |= 1u;
    }
    *dst = h;
}

size_t trav_size(int dtype) { return dtype == BVHGPU_F32 ? sizeof(TravNode<float>) : sizeof(TravNode<double>); }
size_t scalar_size(int dtype) { return dtype == BVHGPU_F32 ? 4 : 8; }
size_t slot_bytes(int dtype) { return (dtype == BVHGPU_F32 ? TopCfg<float>::SLOTS : TopCfg<double>::SLOTS) * 4; }

struct GroupGuard {   // an exception between ncclGroupStart and ncclGroupEnd must not leave the thread's group open
    bool open = false;
    void start() { BVH_RCCL(rccl().GroupStart()); open = true; }
    void end() { open = false; BVH_RCCL(rccl().GroupEnd()); }
    ~GroupGuard() { if (open) (void)rccl().GroupEnd(); }
};

// The arrays of one broadcast, sized from what every rank knows.  root_src: the root's tree, or NULL when the root has
// nothing valid to send — it then sends scratch of the announced sizes (the header says so), because the peers are already
// committed to receiving.  with_header: the status header travels in the same group (bvhgpu_bcast_known) and the peers
// look at it later (pending_recv); otherwise the caller has read it already and passes what it said.
void bcast_arrays(bvhgpu_comm* c, bvhgpu_tree** trees, int root, int dtype, size_t n, size_t n_trav, bool unfolded, bool tris,
                  bool slots, bool root_valid, bool with_header, bool exact_only) {
    const int ndev = (int)c->comms.size();
    const size_t tb = n_trav * trav_size(dtype), ab = n * 6 * scalar_size(dtype), sb = slots ? slot_bytes(dtype) : 0,
                 gb = tris ? n * 9 * scalar_size(dtype) : 0;
    const int local_root = root - c->first_rank;   // index into trees[] if the root is one of this process's devices
    const bool have_root = local_root >= 0 && local_root < ndev;
    for (int i = 0; i < ndev; i++) {
        bvhgpu_tree* t = trees[i];
        if (i == local_root) continue;
        BVH_HIP(hipSetDevice(c->ctxs[i]->device));
        t->pending_build = false; t->pending_recv = false;
        t->built = false; t->flattened = false; t->lazy_flat = false; t->has_wide = false; t->exact_only = exact_only;
        t->dtype = dtype; t->n = n; t->n_trav = n_trav; t->n_nodes = 0; t->n_flat = 0; t->unfolded = unfolded;
        t->gen++;
        t->trav.reserve(tb + 16);
        t->aabbs.reserve(ab + 16);
        if (sb) t->slot_entry.reserve(sb); else t->slot_entry.release();
        if (gb) t->tris.reserve(gb + 16);
        t->has_tris = gb != 0;
        if (with_header && !t->pin_recv) BVH_HIP(hipHostMalloc(&t->pin_recv, 64, hipHostMallocDefault));
    }
    // send buffers of the root: its tree's arrays, or scratch when it has none to offer
    const void *s_trav = nullptr, *s_aabb = nullptr, *s_slot = nullptr, *s_tris = nullptr;
    if (have_root) {
        bvhgpu_tree* r = trees[local_root];
        if (root_valid) { BVH_HIP(hipSetDevice(c->ctxs[local_root]->device)); ensure_flat_arrays(r); }   // (a lazy flatten: the arrays that travel are written now, on the root's stream)
        if (root_valid) { s_trav = r->trav.p; s_aabb = r->aabbs.p; s_slot = r->slot_entry.p; s_tris = r->tris.p; }
        else {
            bvhgpu_ctx* rc = c->ctxs[local_root];
            BVH_HIP(hipSetDevice(rc->device));
            // all-ones scratch: NaN boxes, exits beyond the array, an empty slot table — a peer that walks what it received before
            // looking at the header (bvhgpu_traverse_async_*) finds a tree that every ray leaves at once, never a cycle
            const size_t sz = std::max(std::max(tb, ab), std::max(sb, gb)) + 16;
            rc->upload.reserve(sz);
            BVH_HIP(hipMemsetAsync(rc->upload.p, 0xFF, sz, rc->stream));
            s_trav = s_aabb = s_slot = s_tris = rc->upload.p;
        }
    }
    GroupGuard grp;
    grp.start();
    for (int i = 0; i < ndev; i++) {
        bvhgpu_tree* t = trees[i];
        BVH_HIP(hipSetDevice(c->ctxs[i]->device));
        hipStream_t st = c->ctxs[i]->stream;
        const bool is_root = i == local_root;
        // (only the root's send buffer is read; a peer passes its receive buffer for both)
        if (with_header) BVH_RCCL(rccl().Broadcast(have_root ? c->hdr_dev[local_root] : c->hdr_dev[i], c->hdr_dev[i], 64, ncclUint8, root, c->comms[i], st));
        if (tb) BVH_RCCL(rccl().Broadcast(have_root ? s_trav : t->trav.p, is_root ? const_cast<void*>(s_trav) : t->trav.p, tb, ncclUint8, root, c->comms[i], st));
        if (ab) BVH_RCCL(rccl().Broadcast(have_root ? s_aabb : t->aabbs.p, is_root ? const_cast<void*>(s_aabb) : t->aabbs.p, ab, ncclUint8, root, c->comms[i], st));
        if (sb) BVH_RCCL(rccl().Broadcast(have_root ? s_slot : t->slot_entry.p, is_root ? const_cast<void*>(s_slot) : t->slot_entry.p, sb, ncclUint8, root, c->comms[i], st));
        if (gb) BVH_RCCL(rccl().Broadcast(have_root ? s_tris : t->tris.p, is_root ? const_cast<void*>(s_tris) : t->tris.p, gb, ncclUint8, root, c->comms[i], st));
    }
    grp.end();
    for (int i = 0; i < ndev; i++) {
        if (i == local_root) continue;
        bvhgpu_tree* t = trees[i];
        BVH_HIP(hipSetDevice(c->ctxs[i]->device));
        t->flattened = true;
        if (with_header) {   // looked at by recv_finalize, when the tree is first waited for
            BVH_HIP(hipMemcpyAsync(t->pin_recv, c->hdr_dev[i], 64, hipMemcpyDeviceToHost, c->ctxs[i]->stream));
            t->pending_recv = true;
            t->recv_comm = c;
        }
        // wide nodes, ordered behind the receive on the stream (optimistic when the header is not known yet: an exact_only tree
        // simply never uses them)
        if (!t->exact_only) { if (dtype == BVHGPU_F32) wide_from_trav<float>(t); else wide_from_trav<double>(t); }
    }
}

// allocate the peers' trees / check what the caller passed.  A peer entry that cannot be received into (built here, another
// ctx's) does not stop the collective: the data lands in a temporary tree and the call reports the mistake afterwards.
struct PeerPlan { std::vector<bool> created; std::vector<bvhgpu_tree*> given; int bad = -1; };
void prepare_trees(bvhgpu_comm* c, bvhgpu_tree** trees, int root, PeerPlan& plan) {
    const int ndev = (int)c->comms.size();
    const int local_root = root - c->first_rank;
    plan.created.assign(ndev, false);
    plan.given.assign(trees, trees + ndev);
    for (int i = 0; i < ndev; i++) {
        if (i == local_root) continue;
        bvhgpu_tree* t = trees[i];
        if (t && t->ctx == c->ctxs[i] && (t->pending_build || t->pending_recv)) {   // whatever was in flight is replaced
            try { BVH_HIP(hipSetDevice(t->ctx->device)); if (t->pending_recv) recv_finalize(t); if (t->pending_build) { if (t->dtype == BVHGPU_F32) build_finalize<float>(t); else build_finalize<double>(t); } }
            catch (...) { t->pending_build = false; t->pending_recv = false; }
        }
        if (t && (t->built || t->ctx != c->ctxs[i])) { plan.bad = i; t = nullptr; }
        if (t && !t->waiters.empty()) settle_waiters(t);   // batches still in flight on the peer's old tree are completed on it first
        if (!t) {
            trees[i] = new bvhgpu_tree();
            trees[i]->ctx = c->ctxs[i];
            plan.created[i] = true;
        }
    }
}
// after the collective: drop what was only created to keep the collective whole
int finish_plan(bvhgpu_comm* c, bvhgpu_tree** trees, PeerPlan& plan, int rc, bool drop_created = false) {
    const int ndev = (int)c->comms.size();
    if (drop_created)
        for (int i = 0; i < ndev; i++) if (plan.created[i]) { bvhgpu_tree_destroy(trees[i]); trees[i] = plan.given[i]; plan.created[i] = false; }
    if (plan.bad >= 0) {
        for (int i = 0; i < ndev; i++)
            if (plan.created[i] && plan.given[i]) { bvhgpu_tree_destroy(trees[i]); trees[i] = plan.given[i]; }
        if (rc == BVHGPU_OK)
            rc = comm_fail(c->ctxs[plan.bad], BVHGPU_INVALID_ARG, "bcast: a peer's tree must be NULL or the result of an earlier bcast / scene import on its ctx (the data was received and dropped)");
    } else if (rc != BVHGPU_OK && rc != BVHGPU_INVALID_ARG) {   // HIP / RCCL failure: nothing usable
        for (int i = 0; i < ndev; i++) if (plan.created[i]) { bvhgpu_tree_destroy(trees[i]); trees[i] = nullptr; }
    }
    return rc;
}

// what the root can say about its tree without touching the device; fills the header.  optimistic: the build is still in
// flight and its outcome comes from the device-side status word.
int root_check(bvhgpu_comm* c, bvhgpu_tree* r, int local_root, bool known, int dtype, size_t n_shapes, unsigned what, BcastHeader& h,
               bool& optimistic) {
    bvhgpu_ctx* rctx = c->ctxs[local_root];
    optimistic = false;
    if (!r) return comm_fail(rctx, BVHGPU_INVALID_ARG, "bcast: the root's tree is NULL");
    if (r->ctx != rctx) return comm_fail(rctx, BVHGPU_INVALID_ARG, "bcast: the root's tree belongs to another ctx");
    if (r->pending_recv) {
        const int rc = comm_guarded(rctx, [&] { BVH_HIP(hipSetDevice(rctx->device)); recv_finalize(r); return (int)BVHGPU_OK; });
        if (rc != BVHGPU_OK) return rc;
    }
    if (r->pending_build) {
        // bvhgpu_rebuild_flat_async + bvhgpu_bcast_known: no host round trip — the header takes the build's outcome from HBM
        if (known && r->pend_flatten && r->dtype == dtype && r->n == n_shapes && r->bstat.p) optimistic = true;
        else {
            const int rc = comm_guarded(rctx, [&] {
                BVH_HIP(hipSetDevice(rctx->device));
                if (r->dtype == BVHGPU_F32) build_finalize<float>(r); else build_finalize<double>(r);
                return (int)BVHGPU_OK;
            });
            if (rc != BVHGPU_OK) return rc;
        }
    }
    if (!r->flattened) return comm_fail(rctx, BVHGPU_NOT_FLATTENED, "bcast: flatten the root's tree first");
    if (known && (r->dtype != dtype || r->n != n_shapes || r->unfolded || !r->slot_entry.p || ((what & BVHGPU_BCAST_TRIANGLES) && !r->has_tris)))
        return comm_fail(rctx, BVHGPU_INVALID_ARG, "bcast_known: the root's tree is not what the call announces (dtype, shape count, triangles; not an uploaded FlatBvh)");
    h.dtype = (uint32_t)r->dtype; h.n = r->n; h.n_trav = r->n_trav;
    h.unfolded = r->unfolded ? 1u : 0u; h.has_tris = r->has_tris ? 1u : 0u; h.has_slots = r->slot_entry.p ? 1u : 0u;
    h.flags = (!optimistic && r->exact_only) ? 1u : 0u;
    return BVHGPU_OK;
}

}  // namespace

namespace bvhgpu {
// The tree received a broadcast on its stream; its header sits in pinned memory once the stream has got that far.
void recv_finalize(bvhgpu_tree* t) {
    if (!t->pending_recv) return;
    BVH_HIP(hipStreamSynchronize(t->ctx->stream));
    t->pending_recv = false;
    const BcastHeader h = *reinterpret_cast<const BcastHeader*>(t->pin_recv);
    // (nothing usable arrived: batches that were enqueued on this generation meanwhile return the same status from their own wait)
    auto nothing = [&](hipError_t e, const char* what, int line) { t->flattened = false; t->failed_gen = t->gen; t->failed_what = what; throw HipFail{e, what, line}; };
    if (h.magic != BCAST_MAGIC || h.status > BH_UNFINISHED) nothing(hipErrorUnknown, "RECV_GARBLED", __LINE__);
    if (h.status == BH_INVALID) nothing(hipErrorInvalidValue, "RECV_INVALID", __LINE__);
    if (h.status == BH_UNFINISHED) nothing(hipErrorNotReady, "RECV_REBROADCAST", __LINE__);
    t->exact_only = (h.flags & 1u) != 0;   // batches that were enqueued meanwhile and walked wide are replayed by their wait
    if (t->exact_only) t->has_wide = false;
}
}  // namespace bvhgpu

extern "C" {

int bvhgpu_comm_unique_id(void* id_out) {
    if (!id_out) return BVHGPU_INVALID_ARG;
    static_assert(BVHGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!rccl().ok()) return BVHGPU_RCCL_ERROR;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return BVHGPU_RCCL_ERROR;
    std::memcpy(id_out, &id, sizeof id);
    return BVHGPU_OK;
}

int bvhgpu_comm_init_rank(bvhgpu_ctx* ctx, int nranks, int rank, const void* id, bvhgpu_comm** out) {
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return comm_fail(ctx, BVHGPU_INVALID_ARG, "comm_init_rank: bad argument");
    *out = nullptr;
    { const int rc = need_rccl(ctx); if (rc != BVHGPU_OK) return rc; }
    bvhgpu_comm* c = new bvhgpu_comm();
    int rc = comm_guarded(ctx, [&] {
        BVH_HIP(hipSetDevice(ctx->device));
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        ncclComm_t comm;
        BVH_RCCL(rccl().CommInitRank(&comm, nranks, uid, rank));
        c->comms.push_back(comm);
        c->ctxs.push_back(ctx);
        void* hd = nullptr;
        BVH_HIP(hipMalloc(&hd, 64));
        c->hdr_dev.push_back(hd);
        c->nranks = nranks; c->first_rank = rank;
        return (int)BVHGPU_OK;
    });
    if (rc != BVHGPU_OK) { bvhgpu_comm_destroy(c); return rc; }
    *out = c;
    return BVHGPU_OK;
}

int bvhgpu_comm_init_all(bvhgpu_ctx* const* ctxs, int ndev, bvhgpu_comm** out) {
    if (!ctxs || !out || ndev < 1) return BVHGPU_INVALID_ARG;
    for (int i = 0; i < ndev; i++) if (!ctxs[i]) return BVHGPU_INVALID_ARG;
    *out = nullptr;
    { const int rc = need_rccl(ctxs[0]); if (rc != BVHGPU_OK) return rc; }
    bvhgpu_comm* c = new bvhgpu_comm();
    int rc = comm_guarded(ctxs[0], [&] {
        std::vector<int> devs(ndev);
        // (BVHGPU_RCCL_SHARED_DEVICE: the tests' stand-in library lets several ranks share the one GPU of a test box)
        const bool shared_ok = std::getenv("BVHGPU_RCCL_SHARED_DEVICE") != nullptr && std::getenv("BVHGPU_RCCL_LIB") != nullptr;
        for (int i = 0; i < ndev; i++) {
            devs[i] = ctxs[i]->device;
            for (int j = 0; j < i && !shared_ok; j++)
                if (devs[j] == devs[i]) return comm_fail(ctxs[0], BVHGPU_INVALID_ARG, "comm_init_all: two ctxs on one device (RCCL wants one rank per GPU)");
        }
        c->comms.resize(ndev);
        BVH_RCCL(rccl().CommInitAll(c->comms.data(), ndev, devs.data()));
        for (int i = 0; i < ndev; i++) {
            c->ctxs.push_back(ctxs[i]);
            BVH_HIP(hipSetDevice(devs[i]));
            void* hd = nullptr;
            BVH_HIP(hipMalloc(&hd, 64));
            c->hdr_dev.push_back(hd);
        }
        c->nranks = ndev; c->first_rank = 0;
        return (int)BVHGPU_OK;
    });
    if (rc != BVHGPU_OK) { bvhgpu_comm_destroy(c); return rc; }
    *out = c;
    return BVHGPU_OK;
}

void bvhgpu_comm_destroy(bvhgpu_comm* c) {
    if (!c) return;
    for (size_t i = 0; i < c->comms.size(); i++) {
        if (i < c->ctxs.size()) { (void)hipSetDevice(c->ctxs[i]->device); (void)hipStreamSynchronize(c->ctxs[i]->stream); }
        if (c->comms[i] && rccl().ok()) (void)rccl().CommDestroy(c->comms[i]);
    }
    for (size_t i = 0; i < c->hdr_dev.size(); i++) {
        if (i < c->ctxs.size()) (void)hipSetDevice(c->ctxs[i]->device);
        if (c->hdr_dev[i]) (void)hipFree(c->hdr_dev[i]);
    }
    delete c;
}

int bvhgpu_comm_info(const bvhgpu_comm* c, int* nranks, int* first_rank, int* n_local) {
    if (!c) return BVHGPU_INVALID_ARG;
    if (nranks) *nranks = c->nranks;
    if (first_rank) *first_rank = c->first_rank;
    if (n_local) *n_local = (int)c->comms.size();
    return BVHGPU_OK;
}

// Which RCCL the broadcasts go through: version code (ncclGetVersion: major * 10000 + minor * 100 + patch), the file the entry
// points were resolved from and whether that copy was already loaded in the process.  Loads RCCL if nothing has yet.
int bvhgpu_rccl_info(int* version, int* shared_with_process, char* library_path, size_t cap) {
    if (!rccl().ok()) return BVHGPU_RCCL_ERROR;
    if (version) { int v = 0; if (!rccl().GetVersion || rccl().GetVersion(&v) != ncclSuccess) v = 0; *version = v; }
    if (shared_with_process) *shared_with_process = rccl().shared ? 1 : 0;
    if (library_path && cap) { std::snprintf(library_path, cap, "%s", rccl().path.c_str()); }
    return BVHGPU_OK;
}

// Every rank knows the scene's type and size (a frame loop over a scene of constant shape count): ONE group of broadcasts,
// enqueued on the streams, no host round trip on any rank — also when the root's build is still in flight.
int bvhgpu_bcast_known(bvhgpu_comm* c, bvhgpu_tree** trees, int root, int dtype, size_t n_shapes, unsigned what) {
    // (argument errors every rank sees alike: returning before the collective leaves nobody waiting)
    if (!c || !trees || root < 0 || root >= c->nranks) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx0 = c->ctxs[0];
    if (dtype != BVHGPU_F32 && dtype != BVHGPU_F64) return comm_fail(ctx0, BVHGPU_INVALID_ARG, "bcast: bad dtype");
    const int local_root = root - c->first_rank;
    const int ndev = (int)c->comms.size();
    const bool have_root = local_root >= 0 && local_root < ndev;
    const size_t n_trav = n_shapes >= 2 ? 2 * n_shapes - 2 : n_shapes;
    BcastHeader h;
    std::memset(&h, 0, sizeof h);
    h.magic = BCAST_MAGIC; h.dtype = (uint32_t)dtype; h.n = n_shapes; h.n_trav = n_trav;
    int root_rc = BVHGPU_OK;
    bool optimistic = false;
    bvhgpu_tree* r = have_root ? trees[local_root] : nullptr;
    if (have_root) {
        root_rc = root_check(c, r, local_root, true, dtype, n_shapes, what, h, optimistic);
        if (root_rc != BVHGPU_OK) { h.status = BH_INVALID; h.dtype = (uint32_t)dtype; h.n = n_shapes; h.n_trav = n_trav; }
    }
    const std::string root_err = have_root && root_rc != BVHGPU_OK ? c->ctxs[local_root]->err : std::string();
    PeerPlan plan;
    prepare_trees(c, trees, root, plan);
    int rc = comm_guarded(ctx0, [&] {
        if (have_root) {
            bvhgpu_ctx* rctx = c->ctxs[local_root];
            BVH_HIP(hipSetDevice(rctx->device));
            hipLaunchKernelGGL(k_bcast_header, dim3(1), dim3(64), 0, rctx->stream, reinterpret_cast<BcastHeader*>(c->hdr_dev[local_root]), h,
                               (optimistic && root_rc == BVHGPU_OK) ? r->bstat.as<uint32_t>() : (const uint32_t*)nullptr);
            if (optimistic && root_rc == BVHGPU_OK) r->bcast_gen = r->gen;   // (a broadcast that carried nothing valid is not "this generation was sent")
        }
        bcast_arrays(c, trees, root, dtype, n_shapes, n_trav, false, (what & BVHGPU_BCAST_TRIANGLES) != 0, true, root_rc == BVHGPU_OK, true, false);
        return (int)BVHGPU_OK;
    });
    rc = finish_plan(c, trees, plan, rc);
    if (root_rc != BVHGPU_OK) { c->ctxs[local_root]->err = root_err; return root_rc; }
    return rc;
}

// The peers know nothing: the 64-byte header travels first (one host round trip per rank), then the arrays.
int bvhgpu_bcast(bvhgpu_comm* c, bvhgpu_tree** trees, int root) {
    if (!c || !trees || root < 0 || root >= c->nranks) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx0 = c->ctxs[0];
    const int local_root = root - c->first_rank;
    const int ndev = (int)c->comms.size();
    const bool have_root = local_root >= 0 && local_root < ndev;
    BcastHeader h;
    std::memset(&h, 0, sizeof h);
    h.magic = BCAST_MAGIC;
    int root_rc = BVHGPU_OK;
    bool optimistic = false;
    if (have_root) {
        root_rc = root_check(c, trees[local_root], local_root, false, 0, 0, 0u, h, optimistic);
        if (root_rc != BVHGPU_OK) h.status = BH_INVALID;
    }
    const std::string root_err = have_root && root_rc != BVHGPU_OK ? c->ctxs[local_root]->err : std::string();
    PeerPlan plan;
    prepare_trees(c, trees, root, plan);
    int rc = comm_guarded(ctx0, [&] {
        if (have_root) {
            bvhgpu_ctx* rctx = c->ctxs[local_root];
            BVH_HIP(hipSetDevice(rctx->device));
            hipLaunchKernelGGL(k_bcast_header, dim3(1), dim3(64), 0, rctx->stream, reinterpret_cast<BcastHeader*>(c->hdr_dev[local_root]), h,
                               (const uint32_t*)nullptr);
        }
        {
            GroupGuard grp;
            grp.start();
            for (int i = 0; i < ndev; i++) {
                BVH_HIP(hipSetDevice(c->ctxs[i]->device));
                const void* send = have_root ? c->hdr_dev[local_root] : c->hdr_dev[i];
                BVH_RCCL(rccl().Broadcast(send, c->hdr_dev[i], sizeof h, ncclUint8, root, c->comms[i], c->ctxs[i]->stream));
            }
            grp.end();
        }
        // any local device's copy will do
        BVH_HIP(hipSetDevice(c->ctxs[0]->device));
        BVH_HIP(hipMemcpyAsync(&h, c->hdr_dev[0], sizeof h, hipMemcpyDeviceToHost, c->ctxs[0]->stream));
        BVH_HIP(hipStreamSynchronize(c->ctxs[0]->stream));
        for (int i = 1; i < ndev; i++) { BVH_HIP(hipSetDevice(c->ctxs[i]->device)); BVH_HIP(hipStreamSynchronize(c->ctxs[i]->stream)); }
        if (h.magic != BCAST_MAGIC || h.dtype > 1u || h.status > BH_UNFINISHED) return comm_fail(ctx0, BVHGPU_RCCL_ERROR, "bcast: header did not arrive intact");
        if (h.status != BH_GOOD)   // every rank sees the same header: nobody goes on to the arrays
            return comm_fail(ctx0, BVHGPU_INVALID_ARG, "bcast: the root reported that it has no valid tree to send (its own call returned the reason)");
        bcast_arrays(c, trees, root, (int)h.dtype, (size_t)h.n, (size_t)h.n_trav, h.unfolded != 0, h.has_tris != 0, h.has_slots != 0, true, false,
                     (h.flags & 1u) != 0);
        return (int)BVHGPU_OK;
    });
    rc = finish_plan(c, trees, plan, rc, rc == BVHGPU_INVALID_ARG);   // the root had nothing: trees created for the receive are dropped again
    if (root_rc != BVHGPU_OK) { c->ctxs[local_root]->err = root_err; return root_rc; }
    return rc;
}

}  // extern "C"
