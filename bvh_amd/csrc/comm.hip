// comm.hip — the one exchange step of the path on several GPUs: the root's flattened tree reaches the
// peers in RCCL broadcasts over xGMI, straight from the root tree's own HBM buffers into the peers' (no staging blob).
// Two ways to form the communicator, as RCCL offers them:
//   bvhgpu_comm_init_all   one process drives ndev GPUs (ncclCommInitAll; one ctx per device)
//   bvhgpu_comm_init_rank  one process per GPU (ncclCommInitRank; the launcher carries the 128-byte id to the peers)
// What travels: the folded traversal array, the shape AABBs, the binary LDS slot table and, if asked for, the triangle
// vertices.  The wide nodes (30 MB at 120 k triangles) are NOT sent: every peer rebuilds them from the traversal array
// with one kernel (flatten.hip k_wide) — cheaper than moving them over a 150 GB/s link.  Rays never travel: each GPU
// generates / owns its shard, and hit lists stay where they were produced.
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
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
        snprintf(buf, sizeof buf, "%s failed: %s (comm.hip line %d)", e.what, ncclGetErrorString(e.err), e.line);
        return comm_fail(ctx, BVHGPU_RCCL_ERROR, buf);
    } catch (const HipFail& e) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s failed: %s (comm.hip line %d)", e.what, hipGetErrorString(e.err), e.line);
        return comm_fail(ctx, e.err == hipErrorOutOfMemory ? BVHGPU_OOM : BVHGPU_HIP_ERROR, buf);
    } catch (const std::bad_alloc&) {
        return comm_fail(ctx, BVHGPU_OOM, "host allocation failed");
    } catch (...) {
        return comm_fail(ctx, BVHGPU_HIP_ERROR, "unknown exception");
    }
}

struct BcastHeader {   // 64 bytes, travels first when the peers do not know the scene's size
    uint32_t magic, dtype;
    uint64_t n, n_trav;
    uint32_t unfolded, has_tris, has_slots, _pad;
    uint64_t _r[3];
};
static_assert(sizeof(BcastHeader) == 64, "bcast header");
constexpr uint32_t BCAST_MAGIC = 0x42564842u;

size_t trav_size(int dtype) { return dtype == BVHGPU_F32 ? sizeof(TravNode<float>) : sizeof(TravNode<double>); }
size_t scalar_size(int dtype) { return dtype == BVHGPU_F32 ? 4 : 8; }
size_t slot_bytes(int dtype) { return (dtype == BVHGPU_F32 ? TopCfg<float>::SLOTS : TopCfg<double>::SLOTS) * 4; }

// the arrays of one broadcast, sized from what every rank knows
void bcast_arrays(bvhgpu_comm* c, bvhgpu_tree** trees, int root, int dtype, size_t n, size_t n_trav, bool unfolded, bool tris,
                  bool slots) {
    const int ndev = (int)c->comms.size();
    const size_t tb = n_trav * trav_size(dtype), ab = n * 6 * scalar_size(dtype), sb = slots ? slot_bytes(dtype) : 0,
                 gb = tris ? n * 9 * scalar_size(dtype) : 0;
    const int local_root = root - c->first_rank;   // index into trees[] if the root is one of this process's devices
    for (int i = 0; i < ndev; i++) {
        bvhgpu_tree* t = trees[i];
        if (i == local_root) continue;
        BVH_HIP(hipSetDevice(c->ctxs[i]->device));
        t->pending_build = false;
        t->built = false; t->flattened = false; t->has_wide = false; t->exact_only = false;
        t->dtype = dtype; t->n = n; t->n_trav = n_trav; t->n_nodes = 0; t->n_flat = 0; t->unfolded = unfolded;
        t->trav.reserve(tb + 16);
        t->aabbs.reserve(ab + 16);
        if (sb) t->slot_entry.reserve(sb); else t->slot_entry.release();
        if (gb) t->tris.reserve(gb + 16);
        t->has_tris = gb != 0;
    }
    BVH_RCCL(ncclGroupStart());
    for (int i = 0; i < ndev; i++) {
        bvhgpu_tree* t = trees[i];
        bvhgpu_tree* src = local_root >= 0 && local_root < ndev ? trees[local_root] : t;   // only the root's send buffer is read
        BVH_HIP(hipSetDevice(c->ctxs[i]->device));
        hipStream_t st = c->ctxs[i]->stream;
        if (tb) BVH_RCCL(ncclBroadcast(src->trav.p, t->trav.p, tb, ncclUint8, root, c->comms[i], st));
        if (ab) BVH_RCCL(ncclBroadcast(src->aabbs.p, t->aabbs.p, ab, ncclUint8, root, c->comms[i], st));
        if (sb) BVH_RCCL(ncclBroadcast(src->slot_entry.p, t->slot_entry.p, sb, ncclUint8, root, c->comms[i], st));
        if (gb) BVH_RCCL(ncclBroadcast(src->tris.p, t->tris.p, gb, ncclUint8, root, c->comms[i], st));
    }
    BVH_RCCL(ncclGroupEnd());
    for (int i = 0; i < ndev; i++) {
        if (i == local_root) continue;
        bvhgpu_tree* t = trees[i];
        BVH_HIP(hipSetDevice(c->ctxs[i]->device));
        t->flattened = true;
        if (dtype == BVHGPU_F32) wide_from_trav<float>(t); else wide_from_trav<double>(t);   // ordered behind the receive on the stream
    }
}

// allocate the peers' trees / check what the caller passed
int prepare_trees(bvhgpu_comm* c, bvhgpu_tree** trees, int root, std::vector<bool>& created) {
    const int ndev = (int)c->comms.size();
    const int local_root = root - c->first_rank;
    created.assign(ndev, false);
    for (int i = 0; i < ndev; i++) {
        if (i == local_root) {
            if (!trees[i]) return comm_fail(c->ctxs[i], BVHGPU_INVALID_ARG, "bcast: the root's tree is NULL");
            if (trees[i]->ctx != c->ctxs[i]) return comm_fail(c->ctxs[i], BVHGPU_INVALID_ARG, "bcast: the root's tree belongs to another ctx");
            continue;
        }
        if (trees[i]) {
            if (trees[i]->built || trees[i]->ctx != c->ctxs[i])
                return comm_fail(c->ctxs[i], BVHGPU_INVALID_ARG, "bcast: a peer's tree must be NULL or the result of an earlier bcast / scene import on its ctx");
        } else {
            trees[i] = new bvhgpu_tree();
            trees[i]->ctx = c->ctxs[i];
            created[i] = true;
        }
    }
    return BVHGPU_OK;
}

}  // namespace

extern "C" {

int bvhgpu_comm_unique_id(void* id_out) {
    if (!id_out) return BVHGPU_INVALID_ARG;
    static_assert(BVHGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return BVHGPU_RCCL_ERROR;
    std::memcpy(id_out, &id, sizeof id);
    return BVHGPU_OK;
}

int bvhgpu_comm_init_rank(bvhgpu_ctx* ctx, int nranks, int rank, const void* id, bvhgpu_comm** out) {
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return comm_fail(ctx, BVHGPU_INVALID_ARG, "comm_init_rank: bad argument");
    *out = nullptr;
    bvhgpu_comm* c = new bvhgpu_comm();
    int rc = comm_guarded(ctx, [&] {
        BVH_HIP(hipSetDevice(ctx->device));
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        ncclComm_t comm;
        BVH_RCCL(ncclCommInitRank(&comm, nranks, uid, rank));
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
    bvhgpu_comm* c = new bvhgpu_comm();
    int rc = comm_guarded(ctxs[0], [&] {
        std::vector<int> devs(ndev);
        for (int i = 0; i < ndev; i++) {
            devs[i] = ctxs[i]->device;
            for (int j = 0; j < i; j++)
                if (devs[j] == devs[i]) return comm_fail(ctxs[0], BVHGPU_INVALID_ARG, "comm_init_all: two ctxs on one device (RCCL wants one rank per GPU)");
        }
        c->comms.resize(ndev);
        BVH_RCCL(ncclCommInitAll(c->comms.data(), ndev, devs.data()));
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
        if (c->comms[i]) (void)ncclCommDestroy(c->comms[i]);
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

// Every rank knows the scene's type and size (a frame loop over a scene of constant shape count): ONE group of broadcasts,
// enqueued on the streams, no host round trip on any rank.
int bvhgpu_bcast_known(bvhgpu_comm* c, bvhgpu_tree** trees, int root, int dtype, size_t n_shapes, unsigned what) {
    if (!c || !trees || root < 0 || root >= c->nranks) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx0 = c->ctxs[0];
    if (dtype != BVHGPU_F32 && dtype != BVHGPU_F64) return comm_fail(ctx0, BVHGPU_INVALID_ARG, "bcast: bad dtype");
    const int local_root = root - c->first_rank;
    const int ndev = (int)c->comms.size();
    if (local_root >= 0 && local_root < ndev) {   // this process holds the root: what it sends must be what the peers expect
        bvhgpu_tree* r = trees[local_root];
        if (!r) return comm_fail(ctx0, BVHGPU_INVALID_ARG, "bcast: the root's tree is NULL");
        if (r->pending_build) {
            int rc = comm_guarded(r->ctx, [&] { BVH_HIP(hipSetDevice(r->ctx->device)); if (r->dtype == BVHGPU_F32) build_finalize<float>(r); else build_finalize<double>(r); return (int)BVHGPU_OK; });
            if (rc != BVHGPU_OK) return rc;
        }
        if (!r->flattened) return comm_fail(r->ctx, BVHGPU_NOT_FLATTENED, "bcast: flatten the root's tree first");
        if (r->dtype != dtype || r->n != n_shapes || r->unfolded || !r->slot_entry.p || ((what & BVHGPU_BCAST_TRIANGLES) && !r->has_tris))
            return comm_fail(r->ctx, BVHGPU_INVALID_ARG, "bcast_known: the root's tree is not what the call announces (dtype, shape count, triangles; not an uploaded FlatBvh)");
    }
    std::vector<bool> created;
    int rc = prepare_trees(c, trees, root, created);
    if (rc != BVHGPU_OK) return rc;
    rc = comm_guarded(ctx0, [&] {
        const size_t n_trav = n_shapes >= 2 ? 2 * n_shapes - 2 : n_shapes;
        bcast_arrays(c, trees, root, dtype, n_shapes, n_trav, false, (what & BVHGPU_BCAST_TRIANGLES) != 0, true);
        return (int)BVHGPU_OK;
    });
    if (rc != BVHGPU_OK)
        for (int i = 0; i < ndev; i++) if (created[i]) { bvhgpu_tree_destroy(trees[i]); trees[i] = nullptr; }
    return rc;
}

// The peers know nothing: a 64-byte header travels first (one host round trip per rank), then the arrays.
int bvhgpu_bcast(bvhgpu_comm* c, bvhgpu_tree** trees, int root) {
    if (!c || !trees || root < 0 || root >= c->nranks) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx0 = c->ctxs[0];
    const int local_root = root - c->first_rank;
    const int ndev = (int)c->comms.size();
    BcastHeader h;
    std::memset(&h, 0, sizeof h);
    if (local_root >= 0 && local_root < ndev) {
        bvhgpu_tree* r = trees[local_root];
        if (!r) return comm_fail(ctx0, BVHGPU_INVALID_ARG, "bcast: the root's tree is NULL");
        if (r->pending_build) {
            int rc = comm_guarded(r->ctx, [&] { BVH_HIP(hipSetDevice(r->ctx->device)); if (r->dtype == BVHGPU_F32) build_finalize<float>(r); else build_finalize<double>(r); return (int)BVHGPU_OK; });
            if (rc != BVHGPU_OK) return rc;
        }
        if (!r->flattened) return comm_fail(r->ctx, BVHGPU_NOT_FLATTENED, "bcast: flatten the root's tree first");
        h.magic = BCAST_MAGIC; h.dtype = (uint32_t)r->dtype; h.n = r->n; h.n_trav = r->n_trav;
        h.unfolded = r->unfolded ? 1u : 0u; h.has_tris = r->has_tris ? 1u : 0u; h.has_slots = r->slot_entry.p ? 1u : 0u;
    }
    std::vector<bool> created;
    int rc = prepare_trees(c, trees, root, created);
    if (rc != BVHGPU_OK) return rc;
    rc = comm_guarded(ctx0, [&] {
        if (local_root >= 0 && local_root < ndev) {
            BVH_HIP(hipSetDevice(c->ctxs[local_root]->device));
            BVH_HIP(hipMemcpyAsync(c->hdr_dev[local_root], &h, sizeof h, hipMemcpyHostToDevice, c->ctxs[local_root]->stream));
            BVH_HIP(hipStreamSynchronize(c->ctxs[local_root]->stream));   // `h` is pageable
        }
        BVH_RCCL(ncclGroupStart());
        for (int i = 0; i < ndev; i++) {
            BVH_HIP(hipSetDevice(c->ctxs[i]->device));
            const void* send = local_root >= 0 && local_root < ndev ? c->hdr_dev[local_root] : c->hdr_dev[i];
            BVH_RCCL(ncclBroadcast(send, c->hdr_dev[i], sizeof h, ncclUint8, root, c->comms[i], c->ctxs[i]->stream));
        }
        BVH_RCCL(ncclGroupEnd());
        // any local device's copy will do
        BVH_HIP(hipSetDevice(c->ctxs[0]->device));
        BVH_HIP(hipMemcpyAsync(&h, c->hdr_dev[0], sizeof h, hipMemcpyDeviceToHost, c->ctxs[0]->stream));
        BVH_HIP(hipStreamSynchronize(c->ctxs[0]->stream));
        for (int i = 1; i < ndev; i++) { BVH_HIP(hipSetDevice(c->ctxs[i]->device)); BVH_HIP(hipStreamSynchronize(c->ctxs[i]->stream)); }
        if (h.magic != BCAST_MAGIC || h.dtype > 1u) return comm_fail(ctx0, BVHGPU_RCCL_ERROR, "bcast: header did not arrive intact");
        bcast_arrays(c, trees, root, (int)h.dtype, (size_t)h.n, (size_t)h.n_trav, h.unfolded != 0, h.has_tris != 0, h.has_slots != 0);
        return (int)BVHGPU_OK;
    });
    if (rc != BVHGPU_OK)
        for (int i = 0; i < ndev; i++) if (created[i]) { bvhgpu_tree_destroy(trees[i]); trees[i] = nullptr; }
    return rc;
}

}  // extern "C"
