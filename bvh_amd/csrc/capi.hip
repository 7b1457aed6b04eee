// capi.hip — the extern "C" boundary declared in include/bvh_mi355x.h.
// Every entry point catches everything, records a message on the ctx and returns a status code.
#include <cstdio>
#include <cstring>
#include <new>

#include "engine.hpp"

using namespace bvhgpu;

namespace {

thread_local std::string g_noctx_err;

int fail(bvhgpu_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_noctx_err = msg;
    return status;
}

template <typename F> int guarded(bvhgpu_ctx* ctx, F&& f) {
    try {
        return f();
    } catch (const HipFail& e) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s failed: %s (line %d)", e.what, hipGetErrorString(e.err), e.line);
        if (e.what && std::strcmp(e.what, "OVERFLOW") == 0) return fail(ctx, BVHGPU_OVERFLOW, "more than 2^32-1 hits in one batch");
        if (e.what && std::strcmp(e.what, "NONFINITE") == 0)
            return fail(ctx, BVHGPU_INVALID_ARG, "shape AABBs contain NaN or infinity (or their centroid extent overflows): the reference panics on "
                                                 "such input (bvh_node.rs:214-217, to_usize().unwrap()); nothing was built");
        if (e.what && std::strcmp(e.what, "REBROADCAST") == 0)
            return fail(ctx, BVHGPU_REBROADCAST, "the tree was broadcast before its build was finalized and the build then needed the slow path (unbalanced tree): "
                                                 "this rank's tree and results are complete, the peers' are not — every rank calls bvhgpu_bcast_known again");
        if (e.what && std::strcmp(e.what, "RECV_REBROADCAST") == 0)
            return fail(ctx, BVHGPU_REBROADCAST, "the root broadcast its tree before the build was finalized and the build was not complete: nothing usable was "
                                                 "received — every rank calls bvhgpu_bcast_known again");
        if (e.what && std::strcmp(e.what, "RECV_INVALID") == 0)
            return fail(ctx, BVHGPU_INVALID_ARG, "the broadcast's root reported that it had no valid tree to send (its own call returned the reason); nothing was received");
        if (e.what && std::strcmp(e.what, "RECV_GARBLED") == 0)
            return fail(ctx, BVHGPU_RCCL_ERROR, "the broadcast header did not arrive intact");
        if (e.what && std::strcmp(e.what, "STALE_TREE") == 0)
            return fail(ctx, BVHGPU_INVALID_ARG, "the tree changed before the asynchronous batch was completed");
        if (e.what && std::strcmp(e.what, "ORDERED_DEPTH") == 0)
            return fail(ctx, BVHGPU_OVERFLOW, "tree deeper than the ordered iterator's 32-entry stack (child_distance_traverse.rs:36)");
        if (e.err == hipErrorOutOfMemory) return fail(ctx, BVHGPU_OOM, buf);
        return fail(ctx, BVHGPU_HIP_ERROR, buf);
    } catch (const std::bad_alloc&) {
        return fail(ctx, BVHGPU_OOM, "host allocation failed");
    } catch (...) {
        return fail(ctx, BVHGPU_HIP_ERROR, "unknown exception");
    }
}

void use_device(bvhgpu_ctx* ctx) { BVH_HIP(hipSetDevice(ctx->device)); }

// bring a caller buffer to HBM: returns a device pointer (either the caller's or the ctx staging copy)
const void* to_device(bvhgpu_ctx* ctx, const void* p, size_t bytes, int mem, DevBuf& stage) {
    if (mem == BVHGPU_DEVICE || bytes == 0) return p;
    stage.reserve(bytes);
    BVH_HIP(hipMemcpyAsync(stage.p, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    return stage.p;
}

void copy_out(bvhgpu_ctx* ctx, void* dst, const void* src_dev, size_t bytes, int mem) {
    if (!bytes) return;
    BVH_HIP(hipMemcpyAsync(dst, src_dev, bytes, mem == BVHGPU_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                           ctx->stream));
    BVH_HIP(hipStreamSynchronize(ctx->stream));
}

void free_tree_buffers(bvhgpu_tree* t) {
    if (t->flat_beside && t->ctx && t->ctx->side) { (void)hipStreamSynchronize(t->ctx->side); t->flat_beside = false; }
    t->aabbs.release(); t->nodes.release(); t->node_start.release(); t->node_count.release();
    t->xbar.release(); t->shape_node.release(); t->flat.release(); t->trav.release(); t->slot_entry.release(); t->node_slot.release(); t->tris.release();
    t->idx[0].release(); t->idx[1].release(); t->bk.release(); t->lvbuf.release();
    t->big[0].release(); t->big[1].release(); t->mid2.release(); t->small.release();
    t->stats[0].release(); t->stats[1].release();
    t->tile_item[0].release(); t->tile_item[1].release(); t->tile_cnt.release(); t->chunk_cnt.release(); t->ctr.release(); t->refit_seg.release();
    t->wide.release(); t->wslot_node.release(); t->wide_guide.release(); t->guide_info.release();
    t->bstat.release();
    if (t->ev_flat0) { (void)hipEventDestroy(t->ev_flat0); t->ev_flat0 = nullptr; }
    if (t->ev_flat) { (void)hipEventDestroy(t->ev_flat); t->ev_flat = nullptr; }
    if (t->ev_top) { (void)hipEventDestroy(t->ev_top); t->ev_top = nullptr; }
    if (t->pin) { (void)hipHostFree(t->pin); t->pin = nullptr; }
    if (t->pin_recv) { (void)hipHostFree(t->pin_recv); t->pin_recv = nullptr; }
}

// an asynchronous build may still be in flight: wait for it and finish / validate it (build.hip build_finalize); likewise a
// broadcast that was received on the stream (comm.hip recv_finalize: its status header tells whether the root's tree was good)
void ensure_built(bvhgpu_tree* t) {
    if (t->pending_recv) recv_finalize(t);
    if (!t->pending_build) return;
    const bool sent_early = t->bcast_gen == t->gen;   // bvhgpu_bcast_known went out before this finalize (optimistic)
    if (t->dtype == BVHGPU_F32) build_finalize<float>(t); else build_finalize<double>(t);
    if (sent_early && t->redone_gen == t->gen) throw HipFail{hipErrorNotReady, "REBROADCAST", __LINE__};
}
int settle(bvhgpu_tree* t) {   // entry points that look at a tree's state first complete its asynchronous build
    if (!t->pending_build && !t->pending_recv) return BVHGPU_OK;
    return guarded(t->ctx, [&] { use_device(t->ctx); ensure_built(t); return (int)BVHGPU_OK; });
}

void detach_waiter(bvhgpu_hits* h) {
    bvhgpu_tree* t = h->wait_tree;
    h->wait_tree = nullptr;
    if (!t) return;
    for (size_t i = 0; i < t->waiters.size(); i++)
        if (t->waiters[i] == h) { t->waiters[i] = t->waiters.back(); t->waiters.pop_back(); break; }
}

// Completes an asynchronous batch (the body of bvhgpu_hits_wait; throws).  Whether the optimistic walk is still good is
// decided from what THIS result object recorded when it was enqueued — the tree's generation and whether that generation was
// still unfinalized — not from the tree's state now: the build may have been finalized since by bvhgpu_tree_wait, by another
// result object's wait, by a flatten / nearest / rebuild call, and a slow-path finalize (`redone_gen`) or a tree that must not be
// walked wide (`exact_only`) invalidates every batch of that generation, whoever notices first.
void finish_hits(bvhgpu_hits* h) {
    if (!h->pend_async) return;
    bvhgpu_ctx* ctx = h->ctx;
    h->pend_async = false;
    detach_waiter(h);
    bvhgpu_tree* t = h->pend_tree;
    BVH_HIP(hipStreamSynchronize(ctx->stream));
    if (!t) return;   // empty batch
    bool rebroadcast = false;
    try { ensure_built(t); }
    catch (const HipFail& e) { if (e.what && std::strcmp(e.what, "REBROADCAST") == 0) rebroadcast = true; else throw; }
    // the generation this batch walked turned out to hold nothing (NaN / inf input, a broadcast whose root had no valid tree) and the
    // error may have been consumed elsewhere — bvhgpu_tree_wait, a rebuild that replaces the tree, another result object's wait: the
    // wait of THIS batch still returns what the synchronous call would have returned, never lists from an unbuilt tree
    if (t->failed_gen != 0 && t->failed_gen == h->pend_gen) throw HipFail{hipErrorInvalidValue, t->failed_what ? t->failed_what : "NONFINITE", __LINE__};
    bool replay = h->pend_on_pending && h->pend_gen == t->gen && (t->redone_gen == h->pend_gen || (h->pend_wide && t->exact_only));
    if (h->pend_gen != t->gen) throw HipFail{hipErrorInvalidValue, "STALE_TREE", __LINE__};   // (rebuild / import settle the waiters first: unreachable)
    const void* rays = h->pend_rays;
    for (;;) {
        if (!replay && traverse_check(h)) break;
        replay = false;
        h->replays++;
        if (h->dtype == BVHGPU_F32) traverse_enqueue<float>(t, static_cast<const bvhgpu_ray_f32*>(rays), h->n_rays, h->flags, h);
        else traverse_enqueue<double>(t, static_cast<const bvhgpu_ray_f64*>(rays), h->n_rays, h->flags, h);
        BVH_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (rebroadcast) throw HipFail{hipErrorNotReady, "REBROADCAST", __LINE__};   // this rank's result is complete; the peers' trees are not
}
// Before a tree's arrays are overwritten or freed (rebuild, refit, import, broadcast receive, destroy): the asynchronous batches
// that were enqueued on it are completed first, so that a replay never runs on the wrong tree and no result object keeps a
// dangling pointer.  A failure is kept in the result object and returned by its own bvhgpu_hits_wait.
void settle_waiters_impl(bvhgpu_tree* t) {
    while (!t->waiters.empty()) {
        bvhgpu_hits* h = t->waiters.back();
        bvhgpu_ctx* hc = h->ctx;
        const std::string keep = hc ? hc->err : std::string();
        const int rc = guarded(hc, [&] { use_device(hc); finish_hits(h); return (int)BVHGPU_OK; });
        if (rc != BVHGPU_OK) { h->deferred_rc = rc; h->deferred_err = hc ? hc->err : std::string(); if (hc) hc->err = keep; }
        if (!t->waiters.empty() && t->waiters.back() == h) { h->wait_tree = nullptr; t->waiters.pop_back(); }   // (defensive: always detached by now)
    }
}

constexpr size_t MAX_SHAPES = (0xFFFFFFFFull - 1) / 3;  // flat indices are u32 (flat_bvh.rs:136)

template <typename T> int do_build(bvhgpu_tree* t, const T* aabbs, size_t n, int mem, bool flat = false, bool async = false) {
    bvhgpu_ctx* ctx = t->ctx;
    if (n && !aabbs) return fail(ctx, BVHGPU_INVALID_ARG, "aabbs is NULL");
    if (n > MAX_SHAPES) return fail(ctx, BVHGPU_OVERFLOW, "too many shapes for u32 flat indices");
    if (mem != BVHGPU_HOST && mem != BVHGPU_DEVICE) return fail(ctx, BVHGPU_INVALID_ARG, "bad mem kind");
    use_device(ctx);
    try { ensure_built(t); }   // a previous asynchronous build of this tree (its outcome no longer matters: everything is rebuilt)
    catch (const HipFail& e) { if (!e.what || (std::strcmp(e.what, "REBROADCAST") != 0 && std::strcmp(e.what, "NONFINITE") != 0 && std::strncmp(e.what, "RECV_", 5) != 0)) throw; }
    settle_waiters_impl(t);
    if (ctx->timing) BVH_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    const T* dev = aabbs;
    if (n && mem == BVHGPU_HOST) {  // upload straight into the tree's own copy
        t->aabbs.reserve(n * 6 * sizeof(T));
        BVH_HIP(hipMemcpyAsync(t->aabbs.p, aabbs, n * 6 * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        dev = t->aabbs.as<T>();
    }
    if (async) build_enqueue<T>(t, dev, n, flat);
    else build_tree<T>(t, dev, n, flat);
    if (flat && n == 0) t->flattened = true;
    if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[1], ctx->stream)); ctx->ev_set |= 1u; }
    return BVHGPU_OK;
}

// the shapes moved: same topology, boxes recomputed (refit.hip)
template <typename T> int do_refit(bvhgpu_tree* t, const T* aabbs, size_t n, int mem) {
    bvhgpu_ctx* ctx = t->ctx;
    use_device(ctx);
    ensure_built(t);
    settle_waiters_impl(t);
    if (!t->built) return fail(ctx, BVHGPU_INVALID_ARG, "refit needs a tree that was built here (imported scenes carry no BvhNode array)");
    join_flat(t);
    if (n != t->n) return fail(ctx, BVHGPU_INVALID_ARG, "refit: the number of shapes differs from the tree's (build again)");
    if (n && !aabbs) return fail(ctx, BVHGPU_INVALID_ARG, "aabbs is NULL");
    if (mem != BVHGPU_HOST && mem != BVHGPU_DEVICE) return fail(ctx, BVHGPU_INVALID_ARG, "bad mem kind");
    use_device(ctx);
    const T* dev = aabbs;
    if (n && mem == BVHGPU_HOST) {  // upload straight into the tree's own copy
        BVH_HIP(hipMemcpyAsync(t->aabbs.p, aabbs, n * 6 * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        dev = t->aabbs.as<T>();
    }
    refit_tree<T>(t, dev);
    return BVHGPU_OK;
}

template <typename T> int new_build(bvhgpu_ctx* ctx, const T* aabbs, size_t n, int mem, bvhgpu_tree** out, bool flat = false) {
    if (!ctx || !out) return fail(ctx, BVHGPU_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
    bvhgpu_tree* t = new bvhgpu_tree();
    t->ctx = ctx;
    t->dtype = Traits<T>::dtype;
    int rc = guarded(ctx, [&] { return do_build<T>(t, aabbs, n, mem, flat); });
    if (rc != BVHGPU_OK) { free_tree_buffers(t); delete t; return rc; }
    *out = t;
    return BVHGPU_OK;
}

template <typename T>
int do_traverse(bvhgpu_tree* tree, const typename Traits<T>::Ray* rays, size_t n_rays, int mem, unsigned flags,
                bvhgpu_hits** hits, bool async = false) {
    if (!tree) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx = tree->ctx;
    if (!hits) return fail(ctx, BVHGPU_INVALID_ARG, "hits is NULL");
    if (!async) { const int rc = settle(tree); if (rc != BVHGPU_OK) return rc; }
    else if (mem != BVHGPU_DEVICE) return fail(ctx, BVHGPU_INVALID_ARG, "asynchronous traversal takes rays that are resident in HBM");
    if (*hits && (*hits)->pend_async) return fail(ctx, BVHGPU_INVALID_ARG, "the result object still holds an asynchronous batch: call bvhgpu_hits_wait first");
    if (tree->dtype != Traits<T>::dtype) return fail(ctx, BVHGPU_DTYPE_MISMATCH, "tree dtype differs from ray dtype");
    if (!tree->flattened) return fail(ctx, BVHGPU_NOT_FLATTENED, "call bvhgpu_flatten first");
    if (n_rays && !rays) return fail(ctx, BVHGPU_INVALID_ARG, "rays is NULL");
    if (n_rays >= 0xFFFFFFFFull) return fail(ctx, BVHGPU_OVERFLOW, "more than 2^32-2 rays in one batch");
    if ((flags & (BVHGPU_TRAVERSE_TRIANGLES | BVHGPU_TRAVERSE_CLOSEST)) && !tree->has_tris)
        return fail(ctx, BVHGPU_INVALID_ARG, "TRIANGLES / CLOSEST need bvhgpu_tree_set_triangles first");
    if ((flags & BVHGPU_TRAVERSE_T_SLICE) && (flags & (BVHGPU_TRAVERSE_TRIANGLES | BVHGPU_TRAVERSE_CLOSEST)))
        return fail(ctx, BVHGPU_INVALID_ARG, "T_SLICE cannot be combined with TRIANGLES / CLOSEST");
    if (flags & (BVHGPU_TRAVERSE_NEAREST_FIRST | BVHGPU_TRAVERSE_FARTHEST_FIRST)) {
        if (!tree->built && !tree->pending_build) return fail(ctx, BVHGPU_INVALID_ARG, "ordered traversal walks the BvhNode array: the tree must have been built here");
        if ((flags & BVHGPU_TRAVERSE_NEAREST_FIRST) && (flags & BVHGPU_TRAVERSE_FARTHEST_FIRST))
            return fail(ctx, BVHGPU_INVALID_ARG, "NEAREST_FIRST and FARTHEST_FIRST are alternatives");
        if (flags & (BVHGPU_TRAVERSE_T_SLICE | BVHGPU_TRAVERSE_STATS))
            return fail(ctx, BVHGPU_INVALID_ARG, "ordered traversal supports the INDICES, TRIANGLES and CLOSEST outputs only");
    } else if (flags & BVHGPU_TRAVERSE_BEST_FIRST) {
        return fail(ctx, BVHGPU_INVALID_ARG, "BEST_FIRST needs NEAREST_FIRST or FARTHEST_FIRST");
    }
    if ((flags & BVHGPU_TRAVERSE_TRIANGLES) && (flags & BVHGPU_TRAVERSE_CLOSEST))
        return fail(ctx, BVHGPU_INVALID_ARG, "TRIANGLES and CLOSEST are alternatives");
    return guarded(ctx, [&] {
        use_device(ctx);
        bvhgpu_hits* h = *hits;
        if (!h) h = new bvhgpu_hits();
        *hits = h;
        const auto* dev = static_cast<const typename Traits<T>::Ray*>(
            to_device(ctx, rays, n_rays * sizeof(typename Traits<T>::Ray), mem, ctx->upload));
        if (async) {
            h->force_binary = false; h->pend_attempts = 0; h->deferred_rc = 0; h->replays = 0;
            h->pend_gen = tree->gen; h->pend_on_pending = tree->pending_build || tree->pending_recv;
            traverse_enqueue<T>(tree, dev, n_rays, flags, h);
            h->pend_async = true;
            h->wait_tree = tree; tree->waiters.push_back(h);
        } else {
            traverse_batch<T>(tree, dev, n_rays, flags, h);
        }
        return (int)BVHGPU_OK;
    });
}

template <typename T>
int do_rays_new(bvhgpu_ctx* ctx, const T* origins, const T* dirs, size_t n, int mem_in, typename Traits<T>::Ray* out,
                int mem_out) {
    if (!ctx) return BVHGPU_INVALID_ARG;
    if (n && (!origins || !dirs || !out)) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    return guarded(ctx, [&] {
        use_device(ctx);
        using Ray = typename Traits<T>::Ray;
        const size_t vb = n * 3 * sizeof(T);
        const T* o = origins; const T* d = dirs;
        Ray* outd = out;
        size_t need = (mem_in == BVHGPU_HOST ? 2 * vb : 0) + (mem_out == BVHGPU_HOST ? n * sizeof(Ray) : 0);
        ctx->upload.reserve(need + 64);
        char* base = ctx->upload.as<char>();
        if (mem_in == BVHGPU_HOST) {
            BVH_HIP(hipMemcpyAsync(base, origins, vb, hipMemcpyHostToDevice, ctx->stream));
            BVH_HIP(hipMemcpyAsync(base + vb, dirs, vb, hipMemcpyHostToDevice, ctx->stream));
            o = reinterpret_cast<const T*>(base); d = reinterpret_cast<const T*>(base + vb);
            base += 2 * vb;
        }
        if (mem_out == BVHGPU_HOST) outd = reinterpret_cast<Ray*>(base);
        rays_new<T>(ctx, o, d, n, outd);
        if (mem_out == BVHGPU_HOST) copy_out(ctx, out, outd, n * sizeof(Ray), BVHGPU_HOST);
        return (int)BVHGPU_OK;
    });
}

// ---- FlatBvh upload: convert the reference layout to the engine layout on the host (a marshalling
// step, like the AABB gather), keeping nav+leaf pairs un-folded unless their boxes are bit-identical.
template <typename T>
int tree_from_flat(bvhgpu_ctx* ctx, const typename Traits<T>::Flat* flat, size_t n_flat, const T* shape_aabbs,
                          size_t n, bvhgpu_tree** out) {
    if (!ctx || !out) return fail(ctx, BVHGPU_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
    if ((n_flat && !flat) || (n && !shape_aabbs)) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    if (n_flat >= 0xFFFFFFFFull) return fail(ctx, BVHGPU_OVERFLOW, "flat array too long");
    // validate indices; every entry maps 1:1 to a traversal entry (no folding: the uploaded shapes
    // may have moved since the build, flat_bvh.rs:411-418 re-tests shape.aabb())
    std::vector<TravNode<T>> trav(n_flat);
    for (size_t i = 0; i < n_flat; i++) {
        const auto& f = flat[i];
        TravNode<T> tn;
        std::memset(&tn, 0, sizeof tn);
        if (f.entry == NONE) {  // leaf entry
            if (f.shape >= n) return fail(ctx, BVHGPU_INVALID_ARG, "flat leaf refers to a shape out of range");
            if (f.exit != i + 1) return fail(ctx, BVHGPU_INVALID_ARG, "flat leaf exit_index must be index+1");
            const T* sb = shape_aabbs + 6 * (size_t)f.shape;
            for (int k = 0; k < 3; k++) { tn.mn[k] = sb[k]; tn.mx[k] = sb[3 + k]; }
            tn.exit = (uint32_t)(i + 1);
            tn.shape = f.shape;
        } else {
            if (f.entry != i + 1 || f.exit <= i || f.exit > n_flat)
                return fail(ctx, BVHGPU_INVALID_ARG, "flat navigator entry/exit out of range");
            for (int k = 0; k < 3; k++) { tn.mn[k] = f.min[k]; tn.mx[k] = f.max[k]; }
            tn.exit = f.exit;
            tn.shape = TRAV_INNER | SLOT_NONE;
        }
        trav[i] = tn;
    }
    // top-of-tree slots (heap numbers, flatten.hip k_top_slots) for the LDS-resident part of traversal:
    // an entry's first child follows it directly, its second child sits at the first child's exit.
    constexpr uint32_t SLOTS = TopCfg<T>::SLOTS;
    std::vector<uint32_t> slot_entry(SLOTS, NONE);
    {
        struct Frame { size_t index; uint32_t exit, heap, kids; };
        std::vector<Frame> stack;
        std::vector<uint32_t> heap(n_flat, SLOT_NONE);
        stack.push_back(Frame{(size_t)-1, (uint32_t)n_flat, 1u, 0u});
        bool binary = true;
        for (size_t i = 0; i < n_flat && binary; i++) {
            while (stack.size() > 1 && stack.back().exit <= i) stack.pop_back();
            Frame& par = stack.back();
            if (++par.kids > 2) { binary = false; break; }
            const uint32_t h = std::min<uint32_t>(2u * par.heap + (par.kids - 1u), SLOT_NONE);
            heap[i] = h;
            if (flat[i].entry != NONE) stack.push_back(Frame{i, flat[i].exit, h, 0u});
        }
        // FlatBvh::flatten only ever emits binary trees; an array that passes the index checks above but is not
        // binary would leave the LDS slot table empty while traversal still starts at slot 2
        if (!binary) return fail(ctx, BVHGPU_INVALID_ARG, "flat array is not a flattened binary tree (an entry has more than two children)");
        if (binary) {
            for (size_t i = 0; i < n_flat; i++) {
                if (heap[i] < SLOTS) slot_entry[heap[i]] = (uint32_t)i;
                if (flat[i].entry != NONE) {
                    const uint32_t ex = flat[i].exit;
                    const uint32_t es = (ex < n_flat && heap[ex] < SLOTS) ? heap[ex] : SLOT_NONE;
                    trav[i].shape = TRAV_INNER | es;
                }
            }
        }
    }
    bvhgpu_tree* t = new bvhgpu_tree();
    t->ctx = ctx; t->dtype = Traits<T>::dtype; t->n = n; t->n_nodes = 0; t->n_flat = n_flat; t->n_trav = n_flat;
    int rc = guarded(ctx, [&] {
        use_device(ctx);
        t->aabbs.reserve(n * 6 * sizeof(T) + 16);
        t->trav.reserve(n_flat * sizeof(TravNode<T>) + 16);
        t->flat.reserve(n_flat * sizeof(typename Traits<T>::Flat) + 16);
        t->slot_entry.reserve(SLOTS * 4);
        BVH_HIP(hipMemcpyAsync(t->slot_entry.p, slot_entry.data(), SLOTS * 4, hipMemcpyHostToDevice, ctx->stream));
        if (n) BVH_HIP(hipMemcpyAsync(t->aabbs.p, shape_aabbs, n * 6 * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        if (n_flat) {
            BVH_HIP(hipMemcpyAsync(t->trav.p, trav.data(), n_flat * sizeof(TravNode<T>), hipMemcpyHostToDevice, ctx->stream));
            BVH_HIP(hipMemcpyAsync(t->flat.p, flat, n_flat * sizeof(typename Traits<T>::Flat), hipMemcpyHostToDevice, ctx->stream));
        }
        BVH_HIP(hipStreamSynchronize(ctx->stream));
        return (int)BVHGPU_OK;
    });
    if (rc != BVHGPU_OK) { free_tree_buffers(t); delete t; return rc; }
    t->built = false; t->flattened = true; t->unfolded = true; t->lazy_flat = false;
    *out = t;
    return BVHGPU_OK;
}

template <typename T>
int do_pairs(bvhgpu_ctx* ctx, const typename Traits<T>::Ray* rays, const T* tris, size_t n, int mem, T* out) {
    if (!ctx) return BVHGPU_INVALID_ARG;
    if (n && (!rays || !tris || !out)) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    if (n >= 0xFFFFFFFFull) return fail(ctx, BVHGPU_OVERFLOW, "too many pairs in one call");
    return guarded(ctx, [&] {
        use_device(ctx);
        using Ray = typename Traits<T>::Ray;
        const Ray* rd = rays; const T* td = tris; T* od = out;
        if (mem == BVHGPU_HOST) {
            const size_t rb = n * sizeof(Ray), tb = n * 9 * sizeof(T), ob = n * 3 * sizeof(T);
            ctx->upload.reserve(rb + tb + ob + 64);
            char* base = ctx->upload.as<char>();
            BVH_HIP(hipMemcpyAsync(base, rays, rb, hipMemcpyHostToDevice, ctx->stream));
            BVH_HIP(hipMemcpyAsync(base + rb, tris, tb, hipMemcpyHostToDevice, ctx->stream));
            rd = reinterpret_cast<const Ray*>(base); td = reinterpret_cast<const T*>(base + rb); od = reinterpret_cast<T*>(base + rb + tb);
        }
        ray_triangle_pairs<T>(ctx, rd, td, n, od);
        if (mem == BVHGPU_HOST) copy_out(ctx, out, od, n * 3 * sizeof(T), BVHGPU_HOST);
        return (int)BVHGPU_OK;
    });
}

template <typename T>
int do_nearest(bvhgpu_tree* t, const T* points, size_t n, int mem, int kind, uint32_t* out_shape, T* out_dist) {
    if (!t) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx = t->ctx;
    { const int rc = settle(t); if (rc != BVHGPU_OK) return rc; }
    if (t->dtype != Traits<T>::dtype) return fail(ctx, BVHGPU_DTYPE_MISMATCH, "tree dtype differs from point dtype");
    if (!t->flattened) return fail(ctx, BVHGPU_NOT_FLATTENED, "call bvhgpu_flatten first");
    if (n && (!points || !out_shape || !out_dist)) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    if (kind != 0 && kind != 1) return fail(ctx, BVHGPU_INVALID_ARG, "shape kind must be 0 (AABB) or 1 (triangle)");
    if (kind == 1 && !t->has_tris) return fail(ctx, BVHGPU_INVALID_ARG, "triangle distance needs bvhgpu_tree_set_triangles first");
    if (n >= 0xFFFFFFFFull) return fail(ctx, BVHGPU_OVERFLOW, "too many points in one call");
    return guarded(ctx, [&] {
        use_device(ctx);
        const T* pd = points; uint32_t* sd = out_shape; T* dd = out_dist;
        if (mem == BVHGPU_HOST) {
            const size_t pb = n * 3 * sizeof(T), sb = n * 4, db = n * sizeof(T);
            ctx->upload.reserve(pb + sb + db + 64);
            char* base = ctx->upload.as<char>();
            if (pb) BVH_HIP(hipMemcpyAsync(base, points, pb, hipMemcpyHostToDevice, ctx->stream));
            pd = reinterpret_cast<const T*>(base); dd = reinterpret_cast<T*>(base + pb); sd = reinterpret_cast<uint32_t*>(base + pb + db);
        }
        if (t->n == 0) {   // empty hierarchy → None for every query (flat_bvh.rs:518-520)
            if (n) { BVH_HIP(hipMemsetAsync(sd, 0xFF, n * 4, ctx->stream)); BVH_HIP(hipMemsetAsync(dd, 0, n * sizeof(T), ctx->stream)); }
        } else {
            nearest_batch<T>(t, pd, n, kind, sd, dd);
        }
        if (mem == BVHGPU_HOST) { copy_out(ctx, out_dist, dd, n * sizeof(T), BVHGPU_HOST); copy_out(ctx, out_shape, sd, n * 4, BVHGPU_HOST); }
        return (int)BVHGPU_OK;
    });
}


// ---- host-resident batches (ABI 7): bvhgpu_traverse_host_* / bvhgpu_build_traverse_host_* -------------------------------------------
// What GpuBvh::traverse_batch of the Rust shim costs a caller whose rays live in host memory and who wants the hit lists back there is a
// PCIe problem, not a kernel problem: 1 M rays are 36 MB as Ray structs and 24 MB as origins + directions, the CSR offsets 4 MB.  The
// batch is cut into chunks (host_batch_upload says how), each an ordinary asynchronous batch on a result object of its own:
//   upload stream  chunk k+1: origins + directions H2D (the copy engines; one transfer per chunk in the OD6 layout)
//   main stream    [the tree's build, if one is in flight: the uploads do not wait for it]  chunk k: Ray::new (k_rays_new: the correctly
//                  rounded divide and square root give the bits Ray::new gives, ray_impl.rs:70-80), the walk + CSR assembly, then the
//                  chunk's index list appended to the batch's and its offsets rebased into the batch's array — through their host
//                  addresses straight into the caller's arrays when those are pinned (BVHGPU_TUNE_HOST_ZERO_COPY bit 1)
//   download stream (result arrays that are NOT pinned) chunk k-1: offsets D2H; behind the last chunk the index lists, as many entries as
//                  the previous batch had hits
// and ONE host wait at the end.  A chunk that had to be replayed (hit pool too small on a first batch, a build that finished on the
// slow path) sends its results again, in one piece.
void free_host_batch(bvhgpu_ctx* ctx) {
    HostBatch* hb = ctx->host;
    if (!hb) return;
    ctx->host = nullptr;
    for (auto& h : hb->hits) if (h) { bvhgpu_hits_destroy(h); h = nullptr; }
    for (auto& e : hb->ev_up) if (e) (void)hipEventDestroy(e);
    for (auto& e : hb->ev_done) if (e) (void)hipEventDestroy(e);
    if (hb->up) { (void)hipStreamSynchronize(hb->up); (void)hipStreamDestroy(hb->up); }
    if (hb->up2) { (void)hipStreamSynchronize(hb->up2); (void)hipStreamDestroy(hb->up2); }
    for (auto& e : hb->ev_up2) if (e) (void)hipEventDestroy(e);
    if (hb->ev_main) (void)hipEventDestroy(hb->ev_main);
    if (hb->ev_aabbs) (void)hipEventDestroy(hb->ev_aabbs);
    hb->indices.release();
    if (hb->down) { (void)hipStreamSynchronize(hb->down); (void)hipStreamDestroy(hb->down); }
    hb->od.release(); hb->rays.release(); hb->offsets.release();
    delete hb;
}

// the address under which the device sees a host range (pinned: bvhgpu_host_alloc / _register), or NULL for pageable memory
template <typename U> U* dev_visible(U* p) {
    void* d = nullptr;
    if (!p) return nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void*>(static_cast<const void*>(p)), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return static_cast<U*>(d);
}
constexpr unsigned HOST_READ_BLOCKS = 48;   // workgroups of a kernel that reads pinned host memory: 12 K lanes x 24 B in flight cover the link's latency

void host_batch_init(bvhgpu_ctx* ctx) {
    if (!ctx->host) ctx->host = new HostBatch();
    HostBatch& hb = *ctx->host;
    if (!hb.up) BVH_HIP(hipStreamCreateWithFlags(&hb.up, hipStreamNonBlocking));
    if (!hb.up2) BVH_HIP(hipStreamCreateWithFlags(&hb.up2, hipStreamNonBlocking));
    if (!hb.down) BVH_HIP(hipStreamCreateWithFlags(&hb.down, hipStreamNonBlocking));
    if (!hb.ev_main) BVH_HIP(hipEventCreateWithFlags(&hb.ev_main, hipEventDisableTiming));
    if (!hb.ev_aabbs) BVH_HIP(hipEventCreateWithFlags(&hb.ev_aabbs, hipEventDisableTiming));
    // (the upload streams do NOT wait for the main stream: the previous host batch ended with a wait for everything it had enqueued, and
    //  a build that bvhgpu_rebuild_flat_async_* has just put on the main stream is exactly what the ray upload is meant to run beside)
}

// phase A: the batch's chunks and their uploads (up stream).  Nothing here depends on the tree, so the fused entry
// (bvhgpu_build_traverse_host_*) enqueues it right behind the shapes' upload and BEFORE the build's launches: the ray upload — 24 MB
// against the build's 3 — is the long pole.
template <typename T>
void host_batch_upload(bvhgpu_ctx* ctx, const T* origins, const T* directions, size_t n_rays, bool od6) {
    using Ray = typename Traits<T>::Ray;
    HostBatch& hb = *ctx->host;
    if (od6) directions = origins + 3;
    hb.total = 0; hb.fetched = false; hb.n_rays = n_rays; hb.chunks = 0; hb.with_od = directions != nullptr; hb.od6 = od6;
    if (n_rays == 0) return;
    // Chunks: the walks of all chunks run one after the other on the main stream, each as soon as its rays have arrived; what is
    // left when the last upload ends is the LAST chunk's conversion + walk + offsets download, so the last chunk is the small one
    // (an eighth of the batch, at least 64 K rays: below that the walk's persistent workgroups run half empty) and the others share
    // the rest evenly.  Fewer, larger chunks walk faster in total (1 M rays: 123 µs in one walk, 4 x 47 µs in four).
    const int knob = ctx->tune[BVHGPU_TUNE_HOST_CHUNKS];
    const int want = knob & 0xFF;
    const bool two_up = (knob & 0x100) != 0;   // (experiment: origins and directions on two upload streams)
    int K = want > 0 ? std::min(want, (int)HostBatch::MAX_CHUNKS) : (n_rays >= 524288 ? 3 : (n_rays >= 262144 ? 2 : 1));
    K = (int)std::min<size_t>((size_t)K, std::max<size_t>(n_rays / 32768, 1));
    const size_t last = K > 1 ? std::min(std::max<size_t>(n_rays / 8, 65536), n_rays / (size_t)K) : 0;
    const size_t body = n_rays - last;
    for (int k = 0; k <= K; k++) {
        size_t r = k == K ? n_rays : (k == K - 1 && K > 1 ? body : body * (size_t)k / (size_t)std::max(K - 1, 1));
        if (k != K) r = std::min(n_rays, (r + 1023) & ~(size_t)1023);
        hb.r0[k] = r;
    }
    for (int k = 1; k <= K; k++) hb.r0[k] = std::max(hb.r0[k], hb.r0[k - 1]);
    hb.chunks = K;
    hb.rays.reserve(n_rays * sizeof(Ray));
    hb.offsets.reserve((n_rays + 1) * 4);
    Ray* rays_dev = hb.rays.as<Ray>();
    // pinned ray arrays: the device reads them itself — Ray::new (or a plain copy of the caller's Ray structs) straight out of host memory on
    // the upload stream, one small launch per chunk: no staging buffer, no conversion pass on the main stream, none of the copy engines'
    // 10 - 20 µs between two transfers (seven of them per batch)
    const T* o_vis = (ctx->tune[BVHGPU_TUNE_HOST_ZERO_COPY] & 1) ? dev_visible(origins) : nullptr;
    const T* d_vis = (o_vis && directions) ? (od6 ? o_vis + 3 : dev_visible(directions)) : nullptr;
    hb.zero_copy_in = o_vis && (!directions || d_vis);
    if (directions && !hb.zero_copy_in) hb.od.reserve(2 * n_rays * 3 * sizeof(T));
    for (int k = 0; k < K; k++) {
        const size_t a = hb.r0[k], nk = hb.r0[k + 1] - a;
        if (!hb.ev_up[k]) BVH_HIP(hipEventCreateWithFlags(&hb.ev_up[k], hipEventDisableTiming));
        if (!hb.ev_up2[k]) BVH_HIP(hipEventCreateWithFlags(&hb.ev_up2[k], hipEventDisableTiming));
        if (!hb.ev_done[k]) BVH_HIP(hipEventCreateWithFlags(&hb.ev_done[k], hipEventDisableTiming));
        if (nk && hb.zero_copy_in) {
            if (directions) rays_new<T>(ctx, o_vis + (od6 ? 6 : 3) * a, d_vis + (od6 ? 6 : 3) * a, nk, rays_dev + a, hb.up, HOST_READ_BLOCKS, od6 ? 6u : 3u);
            else copy16(hb.up, reinterpret_cast<const Ray*>(o_vis) + a, rays_dev + a, nk * sizeof(Ray));
        } else if (nk) {
            if (od6) {   // origin and direction of a ray side by side: ONE transfer per chunk
                BVH_HIP(hipMemcpyAsync(hb.od.as<T>() + 6 * a, origins + 6 * a, nk * 6 * sizeof(T), hipMemcpyHostToDevice, hb.up));
            } else if (directions) {
                BVH_HIP(hipMemcpyAsync(hb.od.as<T>() + 3 * a, origins + 3 * a, nk * 3 * sizeof(T), hipMemcpyHostToDevice, hb.up));
                BVH_HIP(hipMemcpyAsync(hb.od.as<T>() + 3 * n_rays + 3 * a, directions + 3 * a, nk * 3 * sizeof(T), hipMemcpyHostToDevice, two_up ? hb.up2 : hb.up));
                if (two_up) { BVH_HIP(hipEventRecord(hb.ev_up2[k], hb.up2)); BVH_HIP(hipStreamWaitEvent(hb.up, hb.ev_up2[k], 0)); }
            } else {   // the caller's own Ray structs (36 / 72 bytes per ray), used as they are
                BVH_HIP(hipMemcpyAsync(rays_dev + a, reinterpret_cast<const Ray*>(origins) + a, nk * sizeof(Ray), hipMemcpyHostToDevice, hb.up));
            }
        }
        BVH_HIP(hipEventRecord(hb.ev_up[k], hb.up));
    }
}

// phase B: per chunk, on the main stream (behind the tree's build if one is in flight): Ray::new, the walk + CSR assembly as an ordinary
// asynchronous batch, the offsets rebased into the batch's array and the index list appended to the batch's; the offsets' download on the
// down stream, the index lists' behind the last chunk — as many entries as the PREVIOUS batch had hits (a frame loop's totals change
// little: the rest, if any, follows after the wait); then ONE host wait
template <typename T>
int host_batch_walk(bvhgpu_tree* tree, unsigned flags, uint32_t* offsets, uint32_t* indices, size_t indices_cap, uint64_t* total) {
    using Ray = typename Traits<T>::Ray;
    bvhgpu_ctx* ctx = tree->ctx;
    HostBatch& hb = *ctx->host;
    const size_t n_rays = hb.n_rays;
    const int K = hb.chunks;
    if (n_rays == 0) {   // (the tree's build, if one is in flight, is completed by whoever looks at the tree next)
        offsets[0] = 0;
        return BVHGPU_OK;
    }
    Ray* rays_dev = hb.rays.as<Ray>();
    uint32_t* offs_all = hb.offsets.as<uint32_t>();
    hipStream_t st = ctx->stream;
    // pinned result arrays: the device writes them itself (k_offsets_rebase / k_indices_append store through the host addresses): no download
    const bool zc_out = (ctx->tune[BVHGPU_TUNE_HOST_ZERO_COPY] & 2) != 0;
    hb.offsets_host = zc_out ? dev_visible(offsets) : nullptr;
    hb.indices_host = (hb.offsets_host && indices && indices_cap) ? dev_visible(indices) : nullptr;
    hb.idx_stage = hb.indices_host ? indices_cap : (indices ? std::min(indices_cap, HostBatch::IDX_STAGE_MAX) : 0);
    if (hb.idx_stage && !hb.indices_host) hb.indices.reserve(hb.idx_stage * 4);
    int enq = 0;
    auto finish_all = [&](bool swallow) {   // every chunk that was enqueued is completed, whatever the first failure was
        bool replayed = false;
        for (int k = 0; k < enq; k++) {
            bvhgpu_hits* h = hb.hits[k];
            if (!swallow) { finish_hits(h); replayed = replayed || h->replays != 0; continue; }
            try { finish_hits(h); } catch (...) { h->pend_async = false; detach_waiter(h); }
        }
        return replayed;
    };
    auto rebase_chunk = [&](int k) {
        const size_t a = hb.r0[k], nk = hb.r0[k + 1] - a;
        offsets_rebase(st, hb.hits[k]->offsets.template as<uint32_t>(), nk, offs_all + a, hb.offsets_host ? hb.offsets_host + a : nullptr,
                       hb.hits[k]->indices.template as<uint32_t>(), hb.indices_host ? hb.indices_host : (hb.idx_stage ? hb.indices.as<uint32_t>() : nullptr),
                       hb.idx_stage, k == 0);
    };
    try {
        for (int k = 0; k < K; k++) {
            const size_t a = hb.r0[k], nk = hb.r0[k + 1] - a;
            BVH_HIP(hipStreamWaitEvent(st, hb.ev_up[k], 0));
            if (hb.with_od && !hb.zero_copy_in) {
                if (hb.od6) rays_new<T>(ctx, hb.od.as<T>() + 6 * a, hb.od.as<T>() + 6 * a + 3, nk, rays_dev + a, st, 0, 6);
                else rays_new<T>(ctx, hb.od.as<T>() + 3 * a, hb.od.as<T>() + 3 * n_rays + 3 * a, nk, rays_dev + a, st);
            }
            const int rc = do_traverse<T>(tree, rays_dev + a, nk, BVHGPU_DEVICE, flags, &hb.hits[k], true);
            if (rc != BVHGPU_OK) { const std::string keep = ctx->err; (void)finish_all(true); (void)hipStreamSynchronize(hb.up); ctx->err = keep; return rc; }
            enq = k + 1;
            rebase_chunk(k);
            if (hb.offsets_host) continue;
            BVH_HIP(hipEventRecord(hb.ev_done[k], st));
            BVH_HIP(hipStreamWaitEvent(hb.down, hb.ev_done[k], 0));
            const size_t skip = k ? 1 : 0;   // (offsets[a] of a later chunk is the previous chunk's last entry)
            if (nk + 1 > skip) BVH_HIP(hipMemcpyAsync(offsets + a + skip, offs_all + a + skip, (nk + 1 - skip) * 4, hipMemcpyDeviceToHost, hb.down));
        }
        uint64_t sent = hb.indices_host ? hb.idx_stage : std::min<uint64_t>(hb.guess, hb.idx_stage);
        if (sent && !hb.indices_host) {
            // (behind the last chunk's append on the main stream — also when the offsets went straight to a pinned array and no chunk
            //  has tied the download stream to the main stream yet)
            BVH_HIP(hipEventRecord(hb.ev_done[K - 1], st));
            BVH_HIP(hipStreamWaitEvent(hb.down, hb.ev_done[K - 1], 0));
            BVH_HIP(hipMemcpyAsync(indices, hb.indices.p, sent * 4, hipMemcpyDeviceToHost, hb.down));
        }
        const bool replayed = finish_all(false);
        BVH_HIP(hipStreamSynchronize(hb.down));
        uint64_t tot = 0;
        for (int k = 0; k < K; k++) tot += hb.hits[k]->total;
        if (tot > 0xFFFFFFFFull) throw HipFail{hipErrorInvalidValue, "OVERFLOW", __LINE__};
        if (replayed) {   // what went down came from an incomplete walk: rebase again, one copy
            for (int k = 0; k < K; k++) rebase_chunk(k);
            if (!hb.offsets_host) BVH_HIP(hipMemcpyAsync(offsets, offs_all, (n_rays + 1) * 4, hipMemcpyDeviceToHost, st));
            BVH_HIP(hipStreamSynchronize(st));
            if (!hb.indices_host) sent = 0;
        }
        hb.total = tot;
        hb.guess = tot;
        *total = tot;
        if (indices && tot <= indices_cap) {
            if (hb.indices_host) {
                // (written by the device, all of them: idx_stage == indices_cap >= tot)
            } else if (tot <= hb.idx_stage) {   // the batch's lists are in one piece: whatever the optimistic download did not cover
                if (tot > sent) { BVH_HIP(hipMemcpyAsync(indices + sent, hb.indices.as<uint32_t>() + sent, (tot - sent) * 4, hipMemcpyDeviceToHost, st)); BVH_HIP(hipStreamSynchronize(st)); }
            } else {
                uint64_t base = 0;
                for (int k = 0; k < K; k++) {
                    const uint64_t tk = hb.hits[k]->total;
                    if (tk) BVH_HIP(hipMemcpyAsync(indices + base, hb.hits[k]->indices.p, tk * 4, hipMemcpyDeviceToHost, st));
                    base += tk;
                }
                BVH_HIP(hipStreamSynchronize(st));
            }
            hb.fetched = true;
        }
    } catch (...) {
        (void)finish_all(true);
        (void)hipStreamSynchronize(hb.up); (void)hipStreamSynchronize(hb.up2); (void)hipStreamSynchronize(hb.down);
        throw;
    }
    return BVHGPU_OK;
}

// rebuild == false: the tree as it is (bvhgpu_traverse_host_*).  rebuild: the shapes' AABBs go up first on the upload stream — the
// link is one resource: a second copy beside the rays' would slow both — the rays right behind them, and the main stream builds
// (bvhgpu_rebuild_flat_async_*) as soon as the shapes are there
template <typename T>
int do_traverse_host(bvhgpu_tree* tree, const T* aabbs, size_t n_shapes, bool rebuild, const T* origins, const T* directions, size_t n_rays,
                     unsigned flags, uint32_t* offsets, uint32_t* indices, size_t indices_cap, uint64_t* total) {
    if (!tree) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx = tree->ctx;
    if (!offsets || !total) return fail(ctx, BVHGPU_INVALID_ARG, "offsets / total is NULL");
    if (n_rays && !origins) return fail(ctx, BVHGPU_INVALID_ARG, "origins is NULL");
    if (flags & ~(BVHGPU_TRAVERSE_COHERENT | BVHGPU_TRAVERSE_RAYS_OD6))
        return fail(ctx, BVHGPU_INVALID_ARG, "bvhgpu_traverse_host_* returns index lists: only BVHGPU_TRAVERSE_COHERENT and BVHGPU_TRAVERSE_RAYS_OD6 may be set");
    const bool od6 = (flags & BVHGPU_TRAVERSE_RAYS_OD6) != 0;
    flags &= ~BVHGPU_TRAVERSE_RAYS_OD6;
    if (tree->dtype != Traits<T>::dtype) return fail(ctx, BVHGPU_DTYPE_MISMATCH, "tree dtype differs from ray / shape dtype");
    if (n_rays >= 0xFFFFFFFFull) return fail(ctx, BVHGPU_OVERFLOW, "more than 2^32-2 rays in one batch");
    if (rebuild && n_shapes && !aabbs) return fail(ctx, BVHGPU_INVALID_ARG, "aabbs is NULL");
    if (rebuild && n_shapes > MAX_SHAPES) return fail(ctx, BVHGPU_OVERFLOW, "too many shapes for u32 flat indices");
    *total = 0;
    return guarded(ctx, [&] {
        use_device(ctx);
        if (rebuild) {   // whatever is in flight on the tree is completed before its shape array is overwritten (as bvhgpu_rebuild_* does)
            try { ensure_built(tree); }
            catch (const HipFail& e) { if (!e.what || (std::strcmp(e.what, "REBROADCAST") != 0 && std::strcmp(e.what, "NONFINITE") != 0 && std::strncmp(e.what, "RECV_", 5) != 0)) throw; }
            settle_waiters_impl(tree);
        }
        host_batch_init(ctx);
        HostBatch& hb = *ctx->host;
        if (rebuild && n_shapes) {
            tree->aabbs.reserve(n_shapes * 6 * sizeof(T));
            const T* a_vis = (ctx->tune[BVHGPU_TUNE_HOST_ZERO_COPY] & 1) ? dev_visible(aabbs) : nullptr;
            if (a_vis) copy16(hb.up, a_vis, tree->aabbs.p, n_shapes * 6 * sizeof(T));
            else BVH_HIP(hipMemcpyAsync(tree->aabbs.p, aabbs, n_shapes * 6 * sizeof(T), hipMemcpyHostToDevice, hb.up));
            BVH_HIP(hipEventRecord(hb.ev_aabbs, hb.up));
        }
        host_batch_upload<T>(ctx, origins, directions, n_rays, od6);
        if (rebuild) {
            if (n_shapes) BVH_HIP(hipStreamWaitEvent(ctx->stream, hb.ev_aabbs, 0));
            const int rc = do_build<T>(tree, tree->aabbs.as<T>(), n_shapes, BVHGPU_DEVICE, true, true);
            if (rc != BVHGPU_OK) { (void)hipStreamSynchronize(hb.up); (void)hipStreamSynchronize(hb.up2); return rc; }
        }
        return host_batch_walk<T>(tree, flags, offsets, indices, indices_cap, total);
    });
}

}  // namespace

namespace bvhgpu {
void settle_waiters(bvhgpu_tree* t) { settle_waiters_impl(t); }   // (comm.hip: a peer's tree is about to be overwritten by a broadcast)
}

#ifdef BVH_PROFILE_MID
namespace bvhgpu { void debug_mid_prof(unsigned long long* out, bool reset); }
#endif
#ifdef BVH_LEVEL_PROFILE
namespace bvhgpu { void debug_level_prof(unsigned long long* out, size_t n); }
#endif
#ifdef BVH_SMALL_PROFILE
namespace bvhgpu { void debug_small_prof(unsigned long long* out, size_t n); }
#endif
#ifdef BVH_WIDE_PROFILE
namespace bvhgpu { void debug_wide_prof(unsigned long long* out, size_t n); void debug_wide_util(unsigned long long* out, size_t n); }
#endif

extern "C" {

int bvhgpu_abi_version(void) { return BVHGPU_ABI_VERSION; }

int bvhgpu_device_count(int* out) {
    if (!out) return BVHGPU_INVALID_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out = 0; (void)hipGetLastError(); return BVHGPU_OK; }
    *out = n;
    return BVHGPU_OK;
}

const char* bvhgpu_status_string(int s) {
    switch (s) {
        case BVHGPU_OK: return "ok";
        case BVHGPU_INVALID_ARG: return "invalid argument";
        case BVHGPU_HIP_ERROR: return "HIP error";
        case BVHGPU_OOM: return "out of memory";
        case BVHGPU_OVERFLOW: return "index overflow";
        case BVHGPU_NO_DEVICE: return "no MI355X / HIP device available";
        case BVHGPU_DTYPE_MISMATCH: return "dtype mismatch";
        case BVHGPU_NOT_FLATTENED: return "tree not flattened";
        case BVHGPU_RCCL_ERROR: return "RCCL error";
        case BVHGPU_REBROADCAST: return "broadcast again";
        default: return "unknown status";
    }
}

int bvhgpu_create(int device, void* stream, bvhgpu_ctx** out) {
    if (!out) return BVHGPU_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(nullptr, BVHGPU_NO_DEVICE, "no HIP device"); }
    if (device < 0 || device >= n) return fail(nullptr, BVHGPU_INVALID_ARG, "device index out of range");
    bvhgpu_ctx* ctx = new bvhgpu_ctx();
    ctx->device = device;
    int rc = guarded(ctx, [&] {
        BVH_HIP(hipSetDevice(device));
        if (stream) { ctx->stream = static_cast<hipStream_t>(stream); ctx->own_stream = false; }
        else { BVH_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
        hipDeviceProp_t prop;
        BVH_HIP(hipGetDeviceProperties(&prop, device));
        ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        for (auto& e : ctx->ev) BVH_HIP(hipEventCreate(&e));
        BVH_HIP(hipHostMalloc(&ctx->pinned, 4096, hipHostMallocDefault));
        return (int)BVHGPU_OK;
    });
    if (rc != BVHGPU_OK) { g_noctx_err = ctx->err; delete ctx; return rc; }
    *out = ctx;
    return BVHGPU_OK;
}

void bvhgpu_destroy(bvhgpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_host_batch(ctx);
    ctx->upload.release();
    ctx->counters.release();
    for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* bvhgpu_last_error(const bvhgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : g_noctx_err.c_str(); }

int bvhgpu_synchronize(bvhgpu_ctx* ctx) {
    if (!ctx) return BVHGPU_INVALID_ARG;
    return guarded(ctx, [&] { BVH_HIP(hipStreamSynchronize(ctx->stream)); return (int)BVHGPU_OK; });
}

void* bvhgpu_stream(bvhgpu_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int bvhgpu_device_alloc(bvhgpu_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    *out = nullptr;
    return guarded(ctx, [&] { use_device(ctx); BVH_HIP(hipMalloc(out, bytes ? bytes : 1)); return (int)BVHGPU_OK; });
}
int bvhgpu_device_free(bvhgpu_ctx* ctx, void* p) {
    if (!ctx) return BVHGPU_INVALID_ARG;
    if (!p) return BVHGPU_OK;
    return guarded(ctx, [&] { use_device(ctx); BVH_HIP(hipStreamSynchronize(ctx->stream)); BVH_HIP(hipFree(p)); return (int)BVHGPU_OK; });
}
int bvhgpu_device_copy(bvhgpu_ctx* ctx, void* dst, int dst_mem, const void* src, int src_mem, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    if ((dst_mem != BVHGPU_HOST && dst_mem != BVHGPU_DEVICE) || (src_mem != BVHGPU_HOST && src_mem != BVHGPU_DEVICE))
        return fail(ctx, BVHGPU_INVALID_ARG, "bad mem kind");
    return guarded(ctx, [&] {
        use_device(ctx);
        if (bytes) {
            BVH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));   // (ordered behind the ctx's earlier work)
            BVH_HIP(hipStreamSynchronize(ctx->stream));
        }
        return (int)BVHGPU_OK;
    });
}

// ---- pinned host memory (ABI 7): buffers the DMA engines read and write directly ----
int bvhgpu_host_alloc(bvhgpu_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    *out = nullptr;
    return guarded(ctx, [&] { use_device(ctx); BVH_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault)); return (int)BVHGPU_OK; });
}
int bvhgpu_host_free(bvhgpu_ctx* ctx, void* p) {
    if (!ctx) return BVHGPU_INVALID_ARG;
    if (!p) return BVHGPU_OK;
    return guarded(ctx, [&] { use_device(ctx); BVH_HIP(hipStreamSynchronize(ctx->stream)); BVH_HIP(hipHostFree(p)); return (int)BVHGPU_OK; });
}
int bvhgpu_host_register(bvhgpu_ctx* ctx, void* p, size_t bytes) {
    if (!ctx || !p || !bytes) return fail(ctx, BVHGPU_INVALID_ARG, "NULL / empty range");
    return guarded(ctx, [&] { use_device(ctx); BVH_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault)); return (int)BVHGPU_OK; });
}
int bvhgpu_host_unregister(bvhgpu_ctx* ctx, void* p) {
    if (!ctx || !p) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    return guarded(ctx, [&] { use_device(ctx); BVH_HIP(hipStreamSynchronize(ctx->stream)); BVH_HIP(hipHostUnregister(p)); return (int)BVHGPU_OK; });
}

int bvhgpu_build_f32(bvhgpu_ctx* ctx, const float* aabbs, size_t n, int mem, bvhgpu_tree** out) {
    return new_build<float>(ctx, aabbs, n, mem, out);
}
int bvhgpu_build_f64(bvhgpu_ctx* ctx, const double* aabbs, size_t n, int mem, bvhgpu_tree** out) {
    return new_build<double>(ctx, aabbs, n, mem, out);
}
int bvhgpu_rebuild_f32(bvhgpu_tree* t, const float* aabbs, size_t n, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    if (t->dtype != BVHGPU_F32) return fail(t->ctx, BVHGPU_DTYPE_MISMATCH, "tree is f64");
    return guarded(t->ctx, [&] { return do_build<float>(t, aabbs, n, mem); });
}
int bvhgpu_rebuild_f64(bvhgpu_tree* t, const double* aabbs, size_t n, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    if (t->dtype != BVHGPU_F64) return fail(t->ctx, BVHGPU_DTYPE_MISMATCH, "tree is f32");
    return guarded(t->ctx, [&] { return do_build<double>(t, aabbs, n, mem); });
}

int bvhgpu_refit_f32(bvhgpu_tree* t, const float* aabbs, size_t n, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    if (t->dtype != BVHGPU_F32) return fail(t->ctx, BVHGPU_DTYPE_MISMATCH, "tree is f64");
    return guarded(t->ctx, [&] { return do_refit<float>(t, aabbs, n, mem); });
}
int bvhgpu_refit_f64(bvhgpu_tree* t, const double* aabbs, size_t n, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    if (t->dtype != BVHGPU_F64) return fail(t->ctx, BVHGPU_DTYPE_MISMATCH, "tree is f32");
    return guarded(t->ctx, [&] { return do_refit<double>(t, aabbs, n, mem); });
}

// FlatBvh::build (flat_bvh.rs:328-331) = Bvh::build + flatten in one call: one host round trip instead of two
int bvhgpu_build_flat_f32(bvhgpu_ctx* ctx, const float* aabbs, size_t n, int mem, bvhgpu_tree** out) {
    return new_build<float>(ctx, aabbs, n, mem, out, true);
}
int bvhgpu_build_flat_f64(bvhgpu_ctx* ctx, const double* aabbs, size_t n, int mem, bvhgpu_tree** out) {
    return new_build<double>(ctx, aabbs, n, mem, out, true);
}
int bvhgpu_rebuild_flat_f32(bvhgpu_tree* t, const float* aabbs, size_t n, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    if (t->dtype != BVHGPU_F32) return fail(t->ctx, BVHGPU_DTYPE_MISMATCH, "tree is f64");
    return guarded(t->ctx, [&] { return do_build<float>(t, aabbs, n, mem, true); });
}
int bvhgpu_rebuild_flat_f64(bvhgpu_tree* t, const double* aabbs, size_t n, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    if (t->dtype != BVHGPU_F64) return fail(t->ctx, BVHGPU_DTYPE_MISMATCH, "tree is f32");
    return guarded(t->ctx, [&] { return do_build<double>(t, aabbs, n, mem, true); });
}

// ---- asynchronous step ----
int bvhgpu_rebuild_flat_async_f32(bvhgpu_tree* t, const float* aabbs, size_t n, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    if (t->dtype != BVHGPU_F32) return fail(t->ctx, BVHGPU_DTYPE_MISMATCH, "tree is f64");
    return guarded(t->ctx, [&] { return do_build<float>(t, aabbs, n, mem, true, true); });
}
int bvhgpu_rebuild_flat_async_f64(bvhgpu_tree* t, const double* aabbs, size_t n, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    if (t->dtype != BVHGPU_F64) return fail(t->ctx, BVHGPU_DTYPE_MISMATCH, "tree is f32");
    return guarded(t->ctx, [&] { return do_build<double>(t, aabbs, n, mem, true, true); });
}
int bvhgpu_tree_wait(bvhgpu_tree* t) {
    if (!t) return BVHGPU_INVALID_ARG;
    return settle(t);
}
int bvhgpu_traverse_async_f32(bvhgpu_tree* tree, const bvhgpu_ray_f32* rays, size_t n_rays, int mem, unsigned flags, bvhgpu_hits** hits) {
    return do_traverse<float>(tree, rays, n_rays, mem, flags, hits, true);
}
int bvhgpu_traverse_async_f64(bvhgpu_tree* tree, const bvhgpu_ray_f64* rays, size_t n_rays, int mem, unsigned flags, bvhgpu_hits** hits) {
    return do_traverse<double>(tree, rays, n_rays, mem, flags, hits, true);
}
// Completes an asynchronous batch: waits for the stream, completes the tree's asynchronous build if there is one (input
// validation, unbalanced trees) and replays the batch where the optimistic launch was not enough — a tree that was not
// finished when the walk ran, a tree the wide walk must not be used on, a hit pool / stack / heap that was too small.
int bvhgpu_hits_wait(bvhgpu_hits* h) {
    if (!h || !h->ctx) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx = h->ctx;
    if (!h->pend_async) {   // nothing in flight — or completed meanwhile on behalf of a rebuild / destroy of its tree
        const int rc = h->deferred_rc;
        if (rc != BVHGPU_OK) { ctx->err = h->deferred_err; h->deferred_rc = 0; }
        return rc;
    }
    return guarded(ctx, [&] { use_device(ctx); finish_hits(h); return (int)BVHGPU_OK; });
}

void bvhgpu_tree_destroy(bvhgpu_tree* t) {
    if (!t) return;
    if (t->ctx) { (void)hipSetDevice(t->ctx->device); (void)hipStreamSynchronize(t->ctx->stream); }
    if (!t->waiters.empty()) {   // asynchronous batches still refer to this tree: complete them while it exists
        (void)settle(t);
        settle_waiters_impl(t);
    }
    free_tree_buffers(t);
    delete t;
}

int bvhgpu_tree_info(const bvhgpu_tree* t, int* dtype, size_t* n_shapes, size_t* n_nodes, size_t* n_flat) {
    if (!t) return BVHGPU_INVALID_ARG;
    if (dtype) *dtype = t->dtype;
    if (n_shapes) *n_shapes = t->n;
    if (n_nodes) *n_nodes = t->n_nodes;
    if (n_flat) *n_flat = t->n_flat;
    return BVHGPU_OK;
}

int bvhgpu_tree_nodes(bvhgpu_tree* t, void* out, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    { const int rc = settle(t); if (rc != BVHGPU_OK) return rc; }
    if (!t->built) return fail(t->ctx, BVHGPU_INVALID_ARG, "tree has no node array (imported scene)");
    if (t->n_nodes && !out) return fail(t->ctx, BVHGPU_INVALID_ARG, "out is NULL");
    return guarded(t->ctx, [&] {
        use_device(t->ctx);
        size_t sz = t->dtype == BVHGPU_F32 ? sizeof(bvhgpu_node_f32) : sizeof(bvhgpu_node_f64);
        copy_out(t->ctx, out, t->nodes.p, t->n_nodes * sz, mem);
        return (int)BVHGPU_OK;
    });
}

int bvhgpu_tree_shape_nodes(bvhgpu_tree* t, uint32_t* out, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    { const int rc = settle(t); if (rc != BVHGPU_OK) return rc; }
    if (!t->built) return fail(t->ctx, BVHGPU_INVALID_ARG, "tree has no node array (imported scene)");
    if (t->n && !out) return fail(t->ctx, BVHGPU_INVALID_ARG, "out is NULL");
    return guarded(t->ctx, [&] {
        use_device(t->ctx);
        copy_out(t->ctx, out, t->shape_node.p, t->n * 4, mem);
        return (int)BVHGPU_OK;
    });
}

int bvhgpu_tree_build_levels(const bvhgpu_tree* t, int* levels) {
    if (!t || !levels) return BVHGPU_INVALID_ARG;
    if (t->pending_build) { const int rc = settle(const_cast<bvhgpu_tree*>(t)); if (rc != BVHGPU_OK) return rc; }
    *levels = t->levels;
    return BVHGPU_OK;
}

int bvhgpu_flatten(bvhgpu_tree* t) {
    if (!t) return BVHGPU_INVALID_ARG;
    { const int rc = settle(t); if (rc != BVHGPU_OK) return rc; }
    if (!t->built) return fail(t->ctx, BVHGPU_INVALID_ARG, "tree has no node array (imported scene)");
    bvhgpu_ctx* ctx = t->ctx;
    return guarded(ctx, [&] {
        use_device(ctx);
        if (ctx->timing) BVH_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
        const bool lazy = ctx->tune[BVHGPU_TUNE_FLATTEN_LAZY] != 0;   // (the FlatNode array then follows on first use: bvhgpu_flat_nodes, a binary walk …)
        if (t->dtype == BVHGPU_F32) flatten_tree<float>(t, nullptr, nullptr, 0, nullptr, 0, 0, lazy); else flatten_tree<double>(t, nullptr, nullptr, 0, nullptr, 0, 0, lazy);
        if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[3], ctx->stream)); ctx->ev_set |= 2u; }
        return (int)BVHGPU_OK;
    });
}

int bvhgpu_flat_nodes(bvhgpu_tree* t, void* out, int mem) {
    if (!t) return BVHGPU_INVALID_ARG;
    { const int rc = settle(t); if (rc != BVHGPU_OK) return rc; }
    if (!t->flattened || !t->built) return fail(t->ctx, BVHGPU_NOT_FLATTENED, "call bvhgpu_flatten first");
    if (t->n_flat && !out) return fail(t->ctx, BVHGPU_INVALID_ARG, "out is NULL");
    return guarded(t->ctx, [&] {
        use_device(t->ctx);
        size_t sz = t->dtype == BVHGPU_F32 ? sizeof(bvhgpu_flat_f32) : sizeof(bvhgpu_flat_f64);
        ensure_flat_arrays(t);
        copy_out(t->ctx, out, t->flat.p, t->n_flat * sz, mem);
        return (int)BVHGPU_OK;
    });
}

int bvhgpu_tree_from_flat_f32(bvhgpu_ctx* ctx, const bvhgpu_flat_f32* flat, size_t n_flat, const float* shape_aabbs, size_t n,
                              bvhgpu_tree** out) {
    return tree_from_flat<float>(ctx, flat, n_flat, shape_aabbs, n, out);
}
int bvhgpu_tree_from_flat_f64(bvhgpu_ctx* ctx, const bvhgpu_flat_f64* flat, size_t n_flat, const double* shape_aabbs, size_t n,
                              bvhgpu_tree** out) {
    return tree_from_flat<double>(ctx, flat, n_flat, shape_aabbs, n, out);
}

// ---- scene blob: header | traversal array | shape AABBs | top-of-tree slot table ----
// exact_only: some split of the tree had no SAH winner (empty child bounds, bvh_node.rs:225-230), so a child box is not the join of its
// grandchildren and the importer must not walk it wide (traverse.hip) — it travels with the tree
struct SceneHeader { uint32_t magic, dtype; uint64_t n, n_trav; uint32_t unfolded, exact_only; uint64_t trav_bytes, aabb_bytes, slot_bytes, tri_bytes; };
static constexpr uint32_t SCENE_MAGIC = 0x42564836u;  // "BVH6" (BVH5 + exact_only)
static size_t slot_table_bytes(int dtype) { return (dtype == BVHGPU_F32 ? TopCfg<float>::SLOTS : TopCfg<double>::SLOTS) * 4; }
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int bvhgpu_scene_nbytes(const bvhgpu_tree* t, size_t* nbytes) {
    if (!t || !nbytes) return BVHGPU_INVALID_ARG;
    if (!t->flattened) return BVHGPU_NOT_FLATTENED;
    size_t tsz = t->dtype == BVHGPU_F32 ? sizeof(TravNode<float>) : sizeof(TravNode<double>);
    size_t ssz = t->dtype == BVHGPU_F32 ? 4 : 8;
    *nbytes = 256 + align256(t->n_trav * tsz) + align256(t->n * 6 * ssz) + align256(t->slot_entry.p ? slot_table_bytes(t->dtype) : 0) +
              align256(t->has_tris ? t->n * 9 * ssz : 0);
    return BVHGPU_OK;
}

int bvhgpu_scene_export(bvhgpu_tree* t, void* dst, int mem) {
    if (!t || !dst) return BVHGPU_INVALID_ARG;
    { const int rc = settle(t); if (rc != BVHGPU_OK) return rc; }
    if (!t->flattened) return fail(t->ctx, BVHGPU_NOT_FLATTENED, "call bvhgpu_flatten first");
    bvhgpu_ctx* ctx = t->ctx;
    return guarded(ctx, [&] {
        use_device(ctx);
        ensure_flat_arrays(t);
        size_t tsz = t->dtype == BVHGPU_F32 ? sizeof(TravNode<float>) : sizeof(TravNode<double>);
        size_t ssz = t->dtype == BVHGPU_F32 ? 4 : 8;
        SceneHeader* h = reinterpret_cast<SceneHeader*>(ctx->pinned);
        std::memset(h, 0, 256);
        h->magic = SCENE_MAGIC; h->dtype = (uint32_t)t->dtype; h->n = t->n; h->n_trav = t->n_trav;
        h->unfolded = t->unfolded ? 1u : 0u;
        h->exact_only = t->exact_only ? 1u : 0u;
        h->trav_bytes = t->n_trav * tsz; h->aabb_bytes = t->n * 6 * ssz;
        h->slot_bytes = t->slot_entry.p ? slot_table_bytes(t->dtype) : 0;
        h->tri_bytes = t->has_tris ? t->n * 9 * ssz : 0;
        char* d = static_cast<char*>(dst);
        const hipMemcpyKind kd = mem == BVHGPU_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        BVH_HIP(hipMemcpyAsync(d, h, 256, mem == BVHGPU_DEVICE ? hipMemcpyHostToDevice : hipMemcpyHostToHost, ctx->stream));
        if (h->trav_bytes) BVH_HIP(hipMemcpyAsync(d + 256, t->trav.p, h->trav_bytes, kd, ctx->stream));
        if (h->aabb_bytes) BVH_HIP(hipMemcpyAsync(d + 256 + align256(h->trav_bytes), t->aabbs.p, h->aabb_bytes, kd, ctx->stream));
        if (h->slot_bytes) BVH_HIP(hipMemcpyAsync(d + 256 + align256(h->trav_bytes) + align256(h->aabb_bytes), t->slot_entry.p, h->slot_bytes, kd, ctx->stream));
        if (h->tri_bytes) BVH_HIP(hipMemcpyAsync(d + 256 + align256(h->trav_bytes) + align256(h->aabb_bytes) + align256(h->slot_bytes), t->tris.p, h->tri_bytes, kd, ctx->stream));
        BVH_HIP(hipStreamSynchronize(ctx->stream));  // the pinned header page is reused by the next call
        return (int)BVHGPU_OK;
    });
}

int bvhgpu_scene_import(bvhgpu_ctx* ctx, const void* src, size_t nbytes, int mem, bvhgpu_tree** out) {
    if (!ctx || !out || !src) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    if (nbytes < 256) return fail(ctx, BVHGPU_INVALID_ARG, "scene blob too small");
    bvhgpu_tree* given = *out;
    if (given && given->ctx == ctx && (given->pending_build || given->pending_recv)) (void)settle(given);   // whatever was in flight is replaced
    if (given && (given->built || given->pending_build || given->ctx != ctx))
        return fail(ctx, BVHGPU_INVALID_ARG, "*out must be NULL or a tree from bvhgpu_scene_import on this ctx");
    if (given) settle_waiters_impl(given);
    bvhgpu_tree* t = given ? given : new bvhgpu_tree();
    t->ctx = ctx;
    int rc = guarded(ctx, [&] {
        use_device(ctx);
        SceneHeader* h = reinterpret_cast<SceneHeader*>(ctx->pinned);
        const char* s = static_cast<const char*>(src);
        BVH_HIP(hipMemcpyAsync(h, s, 256, mem == BVHGPU_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyHostToHost, ctx->stream));
        BVH_HIP(hipStreamSynchronize(ctx->stream));
        if (h->magic != SCENE_MAGIC || h->dtype > 1 || h->exact_only > 1 || h->unfolded > 1) return fail(ctx, BVHGPU_INVALID_ARG, "not a bvhgpu scene blob (or one of another ABI version)");
        // the header is untrusted: every section size must be what (n, n_trav, dtype) imply, and the sum must fit the blob
        const SceneHeader hd = *h;   // (the pinned page is reused below)
        const uint64_t tsz = hd.dtype == BVHGPU_F32 ? sizeof(TravNode<float>) : sizeof(TravNode<double>);
        const uint64_t ssz = hd.dtype == BVHGPU_F32 ? 4 : 8;
        if (hd.n > MAX_SHAPES || hd.n_trav > 3 * (uint64_t)MAX_SHAPES) return fail(ctx, BVHGPU_INVALID_ARG, "scene blob: shape / entry count out of range");
        if (hd.trav_bytes != hd.n_trav * tsz || hd.aabb_bytes != hd.n * 6 * ssz || (hd.tri_bytes != 0 && hd.tri_bytes != hd.n * 9 * ssz))
            return fail(ctx, BVHGPU_INVALID_ARG, "scene blob: section sizes do not match the shape / entry counts");
        if (hd.slot_bytes && hd.slot_bytes != slot_table_bytes((int)hd.dtype)) return fail(ctx, BVHGPU_INVALID_ARG, "scene blob slot table size");
        if (!hd.unfolded && hd.n_trav != (hd.n >= 2 ? 2 * hd.n - 2 : hd.n)) return fail(ctx, BVHGPU_INVALID_ARG, "scene blob: entry count does not match the shape count");
        {
            uint64_t need = 256;   // sizes are < 2^40 after the checks above: the sum cannot wrap
            for (uint64_t part : {hd.trav_bytes, hd.aabb_bytes, hd.slot_bytes, hd.tri_bytes}) need += align256((size_t)part);
            if (need > nbytes) return fail(ctx, BVHGPU_INVALID_ARG, "scene blob truncated");
        }
        const size_t tb = hd.trav_bytes, ab = hd.aabb_bytes, sb = hd.slot_bytes, gb = hd.tri_bytes;
        t->flattened = false; t->lazy_flat = false;   // until the new arrays are in place
        t->trav.reserve(tb + 16);
        t->aabbs.reserve(ab + 16);
        const hipMemcpyKind kd = mem == BVHGPU_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        if (tb) BVH_HIP(hipMemcpyAsync(t->trav.p, s + 256, tb, kd, ctx->stream));
        if (ab) BVH_HIP(hipMemcpyAsync(t->aabbs.p, s + 256 + align256(tb), ab, kd, ctx->stream));
        if (sb) { t->slot_entry.reserve(sb); BVH_HIP(hipMemcpyAsync(t->slot_entry.p, s + 256 + align256(tb) + align256(ab), sb, kd, ctx->stream)); }
        else t->slot_entry.release();
        if (gb) { t->tris.reserve(gb); BVH_HIP(hipMemcpyAsync(t->tris.p, s + 256 + align256(tb) + align256(ab) + align256(sb), gb, kd, ctx->stream)); }
        // every copy is enqueued: only now does the tree take the new identity (a failure above leaves a reused tree as it was,
        // apart from buffers that may have grown)
        t->dtype = (int)hd.dtype; t->n = hd.n; t->n_trav = hd.n_trav; t->n_nodes = 0; t->n_flat = 0;
        t->unfolded = hd.unfolded != 0;
        t->has_tris = gb != 0;
        t->built = false; t->flattened = true; t->exact_only = hd.exact_only != 0;
        t->pending_build = false; t->pending_recv = false; t->gen++;
        if (t->exact_only) t->has_wide = false;   // (never walked wide: no wide nodes needed)
        else if (t->dtype == BVHGPU_F32) wide_from_trav<float>(t); else wide_from_trav<double>(t);
        if (mem != BVHGPU_DEVICE) BVH_HIP(hipStreamSynchronize(ctx->stream));
        return (int)BVHGPU_OK;
    });
    if (rc != BVHGPU_OK) { if (!given) { free_tree_buffers(t); delete t; } return rc; }
    *out = t;
    return BVHGPU_OK;
}

// ---- rays ----
int bvhgpu_rays_new_f32(bvhgpu_ctx* ctx, const float* o, const float* d, size_t n, int mem_in, bvhgpu_ray_f32* out, int mem_out) {
    return do_rays_new<float>(ctx, o, d, n, mem_in, out, mem_out);
}
int bvhgpu_rays_new_f64(bvhgpu_ctx* ctx, const double* o, const double* d, size_t n, int mem_in, bvhgpu_ray_f64* out, int mem_out) {
    return do_rays_new<double>(ctx, o, d, n, mem_in, out, mem_out);
}
int bvhgpu_gen_rays_f32(bvhgpu_ctx* ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f32* out_dev) {
    if (!ctx || !bounds || (n && !out_dev)) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    if (n >= 0xFFFFFFFFull) return fail(ctx, BVHGPU_OVERFLOW, "too many rays in one call");
    return guarded(ctx, [&] { use_device(ctx); gen_rays_f32(ctx, first, n, bounds, out_dev); return (int)BVHGPU_OK; });
}
int bvhgpu_gen_rays_f64(bvhgpu_ctx* ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f64* out_dev) {
    if (!ctx || !bounds || (n && !out_dev)) return fail(ctx, BVHGPU_INVALID_ARG, "NULL argument");
    if (n >= 0xFFFFFFFFull) return fail(ctx, BVHGPU_OVERFLOW, "too many rays in one call");
    return guarded(ctx, [&] { use_device(ctx); gen_rays_f64(ctx, first, n, bounds, out_dev); return (int)BVHGPU_OK; });
}

int bvhgpu_gen_primary_rays_f32(bvhgpu_ctx* ctx, const float cam[14], uint32_t width, uint32_t height, uint64_t first, size_t n,
                                bvhgpu_ray_f32* out_dev) {
    if (!ctx || !cam || (n && !out_dev) || !width || !height) return fail(ctx, BVHGPU_INVALID_ARG, "bad argument");
    if (n >= 0xFFFFFFFFull) return fail(ctx, BVHGPU_OVERFLOW, "too many rays in one call");
    return guarded(ctx, [&] { use_device(ctx); gen_primary<float>(ctx, cam, width, height, first, n, out_dev); return (int)BVHGPU_OK; });
}
int bvhgpu_gen_primary_rays_f64(bvhgpu_ctx* ctx, const float cam[14], uint32_t width, uint32_t height, uint64_t first, size_t n,
                                bvhgpu_ray_f64* out_dev) {
    if (!ctx || !cam || (n && !out_dev) || !width || !height) return fail(ctx, BVHGPU_INVALID_ARG, "bad argument");
    if (n >= 0xFFFFFFFFull) return fail(ctx, BVHGPU_OVERFLOW, "too many rays in one call");
    return guarded(ctx, [&] { use_device(ctx); gen_primary<double>(ctx, cam, width, height, first, n, out_dev); return (int)BVHGPU_OK; });
}
int bvhgpu_nearest_f32(bvhgpu_tree* t, const float* points, size_t n, int mem, int kind, uint32_t* out_shape, float* out_dist) {
    return do_nearest<float>(t, points, n, mem, kind, out_shape, out_dist);
}
int bvhgpu_nearest_f64(bvhgpu_tree* t, const double* points, size_t n, int mem, int kind, uint32_t* out_shape, double* out_dist) {
    return do_nearest<double>(t, points, n, mem, kind, out_shape, out_dist);
}
int bvhgpu_ray_triangle_pairs_f32(bvhgpu_ctx* ctx, const bvhgpu_ray_f32* rays, const float* tris, size_t n, int mem, float* out) {
    return do_pairs<float>(ctx, rays, tris, n, mem, out);
}
int bvhgpu_ray_triangle_pairs_f64(bvhgpu_ctx* ctx, const bvhgpu_ray_f64* rays, const double* tris, size_t n, int mem, double* out) {
    return do_pairs<double>(ctx, rays, tris, n, mem, out);
}

// ---- traverse ----
int bvhgpu_traverse_f32(bvhgpu_tree* tree, const bvhgpu_ray_f32* rays, size_t n_rays, int mem, unsigned flags, bvhgpu_hits** hits) {
    return do_traverse<float>(tree, rays, n_rays, mem, flags, hits);
}
int bvhgpu_traverse_f64(bvhgpu_tree* tree, const bvhgpu_ray_f64* rays, size_t n_rays, int mem, unsigned flags, bvhgpu_hits** hits) {
    return do_traverse<double>(tree, rays, n_rays, mem, flags, hits);
}

int bvhgpu_traverse_host_f32(bvhgpu_tree* tree, const float* origins, const float* directions, size_t n_rays, unsigned flags, uint32_t* offsets,
                             uint32_t* indices, size_t indices_cap, uint64_t* total) {
    return do_traverse_host<float>(tree, nullptr, 0, false, origins, directions, n_rays, flags, offsets, indices, indices_cap, total);
}
int bvhgpu_traverse_host_f64(bvhgpu_tree* tree, const double* origins, const double* directions, size_t n_rays, unsigned flags, uint32_t* offsets,
                             uint32_t* indices, size_t indices_cap, uint64_t* total) {
    return do_traverse_host<double>(tree, nullptr, 0, false, origins, directions, n_rays, flags, offsets, indices, indices_cap, total);
}
int bvhgpu_build_traverse_host_f32(bvhgpu_tree* tree, const float* aabbs, size_t n, const float* origins, const float* directions, size_t n_rays,
                                   unsigned flags, uint32_t* offsets, uint32_t* indices, size_t indices_cap, uint64_t* total) {
    return do_traverse_host<float>(tree, aabbs, n, true, origins, directions, n_rays, flags, offsets, indices, indices_cap, total);
}
int bvhgpu_build_traverse_host_f64(bvhgpu_tree* tree, const double* aabbs, size_t n, const double* origins, const double* directions, size_t n_rays,
                                   unsigned flags, uint32_t* offsets, uint32_t* indices, size_t indices_cap, uint64_t* total) {
    return do_traverse_host<double>(tree, aabbs, n, true, origins, directions, n_rays, flags, offsets, indices, indices_cap, total);
}
int bvhgpu_traverse_host_indices(bvhgpu_ctx* ctx, uint32_t* indices, size_t indices_cap) {
    if (!ctx) return BVHGPU_INVALID_ARG;
    HostBatch* hb = ctx->host;
    if (!hb || (hb->n_rays && hb->chunks == 0)) return fail(ctx, BVHGPU_INVALID_ARG, "no completed bvhgpu_traverse_host_* batch on this ctx");
    if (hb->total > indices_cap) return fail(ctx, BVHGPU_INVALID_ARG, "indices_cap is smaller than the batch's hit total");
    if (hb->total && !indices) return fail(ctx, BVHGPU_INVALID_ARG, "indices is NULL");
    return guarded(ctx, [&] {
        use_device(ctx);
        uint64_t base = 0;
        for (int k = 0; k < hb->chunks; k++) {
            const uint64_t tk = hb->hits[k]->total;
            if (tk) BVH_HIP(hipMemcpyAsync(indices + base, hb->hits[k]->indices.p, tk * 4, hipMemcpyDeviceToHost, ctx->stream));
            base += tk;
        }
        BVH_HIP(hipStreamSynchronize(ctx->stream));
        hb->fetched = true;
        return (int)BVHGPU_OK;
    });
}

static int set_triangles(bvhgpu_tree* t, const void* verts, size_t n, int mem, int dtype) {
    if (!t) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx = t->ctx;
    if (t->dtype != dtype) return fail(ctx, BVHGPU_DTYPE_MISMATCH, "triangle dtype differs from tree dtype");
    if (n != t->n) return fail(ctx, BVHGPU_INVALID_ARG, "one triangle per shape is required");
    if (n && !verts) return fail(ctx, BVHGPU_INVALID_ARG, "verts is NULL");
    return guarded(ctx, [&] {
        use_device(ctx);
        const size_t bytes = n * 9 * (dtype == BVHGPU_F32 ? 4 : 8);
        settle_waiters_impl(t);   // (a replay of a batch in flight must see the vertices it was enqueued with)
        t->tris.reserve(bytes + 16);
        if (bytes) {
            BVH_HIP(hipMemcpyAsync(t->tris.p, verts, bytes, mem == BVHGPU_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
            if (mem != BVHGPU_DEVICE) BVH_HIP(hipStreamSynchronize(ctx->stream));
        }
        t->has_tris = true;
        return (int)BVHGPU_OK;
    });
}
int bvhgpu_tree_set_triangles_f32(bvhgpu_tree* t, const float* verts, size_t n, int mem) { return set_triangles(t, verts, n, mem, BVHGPU_F32); }
int bvhgpu_tree_set_triangles_f64(bvhgpu_tree* t, const double* verts, size_t n, int mem) { return set_triangles(t, verts, n, mem, BVHGPU_F64); }

int bvhgpu_hits_fetch_triangles(bvhgpu_hits* h, void* isect, int mem) {
    if (!h || !h->ctx) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx = h->ctx;
    if (h->pend_async) return fail(ctx, BVHGPU_INVALID_ARG, "the result object holds an asynchronous batch that has not been completed: call bvhgpu_hits_wait first");
    if (!(h->flags & BVHGPU_TRAVERSE_TRIANGLES)) return fail(ctx, BVHGPU_INVALID_ARG, "traverse was run without BVHGPU_TRAVERSE_TRIANGLES");
    return guarded(ctx, [&] {
        use_device(ctx);
        if (isect && h->total) copy_out(ctx, isect, h->isect.p, h->total * 3 * (h->dtype == BVHGPU_F32 ? 4 : 8), mem);
        return (int)BVHGPU_OK;
    });
}
int bvhgpu_hits_fetch_closest(bvhgpu_hits* h, void* isect, uint32_t* shape, int mem) {
    if (!h || !h->ctx) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx = h->ctx;
    if (h->pend_async) return fail(ctx, BVHGPU_INVALID_ARG, "the result object holds an asynchronous batch that has not been completed: call bvhgpu_hits_wait first");
    if (!(h->flags & BVHGPU_TRAVERSE_CLOSEST)) return fail(ctx, BVHGPU_INVALID_ARG, "traverse was run without BVHGPU_TRAVERSE_CLOSEST");
    return guarded(ctx, [&] {
        use_device(ctx);
        if (isect && h->n_rays) copy_out(ctx, isect, h->closest.p, h->n_rays * 3 * (h->dtype == BVHGPU_F32 ? 4 : 8), mem);
        if (shape && h->n_rays) copy_out(ctx, shape, h->closest_prim.p, h->n_rays * 4, mem);
        return (int)BVHGPU_OK;
    });
}

int bvhgpu_hits_info(const bvhgpu_hits* h, size_t* n_rays, uint64_t* total, bvhgpu_traverse_stats* stats) {
    if (!h) return BVHGPU_INVALID_ARG;
    if (h->pend_async) return fail(h->ctx, BVHGPU_INVALID_ARG, "the result object holds an asynchronous batch that has not been completed: call bvhgpu_hits_wait first");
    if (n_rays) *n_rays = h->n_rays;
    if (total) *total = h->total;
    if (stats) *stats = h->stats;
    return BVHGPU_OK;
}

int bvhgpu_hits_walk_info(const bvhgpu_hits* h, unsigned* flags) {
    if (!h || !flags) return BVHGPU_INVALID_ARG;
    if (h->pend_async) return fail(h->ctx, BVHGPU_INVALID_ARG, "the result object holds an asynchronous batch that has not been completed: call bvhgpu_hits_wait first");
    *flags = (h->pend_wide ? BVHGPU_WALK_WIDE : 0u) | (h->pend_wide && h->pend_staged ? BVHGPU_WALK_STAGED : 0u) |
             (h->pend_wide && h->pend_rec8 ? BVHGPU_WALK_REC8 : 0u) | (h->pend_wide && h->pend_guide ? BVHGPU_WALK_F64_GUIDE : 0u);
    return BVHGPU_OK;
}

int bvhgpu_hits_walk_kernel(const bvhgpu_hits* h, char* name, size_t cap) {
    if (!h || !name || cap == 0) return BVHGPU_INVALID_ARG;
    std::snprintf(name, cap, "%s", h->walk_kernel.c_str());
    return BVHGPU_OK;
}

int bvhgpu_hits_fetch(bvhgpu_hits* h, uint32_t* offsets, uint32_t* indices, void* tslice, int mem) {
    if (!h || !h->ctx) return BVHGPU_INVALID_ARG;
    bvhgpu_ctx* ctx = h->ctx;
    if (h->pend_async) return fail(ctx, BVHGPU_INVALID_ARG, "the result object holds an asynchronous batch that has not been completed: call bvhgpu_hits_wait first");
    if (tslice && !(h->flags & BVHGPU_TRAVERSE_T_SLICE)) return fail(ctx, BVHGPU_INVALID_ARG, "traverse was run without BVHGPU_TRAVERSE_T_SLICE");
    if (h->flags & BVHGPU_TRAVERSE_CLOSEST) return fail(ctx, BVHGPU_INVALID_ARG, "CLOSEST produces no CSR: use bvhgpu_hits_fetch_closest");
    return guarded(ctx, [&] {
        use_device(ctx);
        if (offsets) copy_out(ctx, offsets, h->offsets.p, (h->n_rays + 1) * 4, mem);
        if (indices && h->total) copy_out(ctx, indices, h->indices.p, h->total * 4, mem);
        if (tslice && h->total) copy_out(ctx, tslice, h->tslice.p, h->total * 2 * (h->dtype == BVHGPU_F32 ? 4 : 8), mem);
        return (int)BVHGPU_OK;
    });
}

int bvhgpu_hits_device(const bvhgpu_hits* h, const uint32_t** offsets, const uint32_t** indices, const void** tslice) {
    if (!h) return BVHGPU_INVALID_ARG;
    if (h->pend_async) return fail(h->ctx, BVHGPU_INVALID_ARG, "the result object holds an asynchronous batch that has not been completed: call bvhgpu_hits_wait first");
    if (h->flags & BVHGPU_TRAVERSE_CLOSEST) return fail(h->ctx, BVHGPU_INVALID_ARG, "CLOSEST produces no CSR");
    if (offsets) *offsets = h->offsets.as<uint32_t>();
    if (indices) *indices = h->indices.as<uint32_t>();
    if (tslice) *tslice = (h->flags & BVHGPU_TRAVERSE_T_SLICE) ? h->tslice.p : nullptr;
    return BVHGPU_OK;
}

void bvhgpu_hits_destroy(bvhgpu_hits* h) {
    if (!h) return;
    if (h->ctx) { (void)hipSetDevice(h->ctx->device); (void)hipStreamSynchronize(h->ctx->stream); }
    detach_waiter(h);   // an asynchronous batch that is never waited for: its tree forgets it
    h->counts.release(); h->offsets.release(); h->pool.release(); h->pool_t.release();
    h->indices.release(); h->tslice.release(); h->blocksums.release(); h->scan_sums.release(); h->ctr.release();
    h->isect.release(); h->closest.release(); h->closest_prim.release(); h->closest_key.release();
    h->heap_dist.release(); h->heap_node.release();
    h->wg_items.release(); h->raybuf.release();
    if (h->ev_items) (void)hipEventDestroy(h->ev_items);
    h->wcounts.release(); h->ray_mask.release(); h->item_cnt.release(); h->wstack.release(); h->ray_items.release(); h->witems.release();
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
}

int bvhgpu_enable_timing(bvhgpu_ctx* ctx, int on) {
    if (!ctx) return BVHGPU_INVALID_ARG;
    ctx->timing = on != 0;
    return BVHGPU_OK;
}
int bvhgpu_last_timings(bvhgpu_ctx* ctx, bvhgpu_timings* out) {
    if (!ctx || !out) return BVHGPU_INVALID_ARG;
    return guarded(ctx, [&] {
        use_device(ctx);
        BVH_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->ev_set & 1u) (void)hipEventElapsedTime(&ctx->last.build_ms, ctx->ev[0], ctx->ev[1]);
        if (ctx->ev_set & 2u) (void)hipEventElapsedTime(&ctx->last.flatten_ms, ctx->ev[2], ctx->ev[3]);
        if (ctx->ev_set & 4u) {
            (void)hipEventElapsedTime(&ctx->last.traverse_kernel_ms, ctx->ev[4], ctx->ev[5]);
            (void)hipEventElapsedTime(&ctx->last.traverse_total_ms, ctx->ev[4], ctx->ev[6]);
        }
        *out = ctx->last;
        return (int)BVHGPU_OK;
    });
}

int bvhgpu_set_tuning(bvhgpu_ctx* ctx, int knob, int value) {
    if (!ctx || knob < 0 || knob >= BVHGPU_TUNE_COUNT) return BVHGPU_INVALID_ARG;
    ctx->tune[knob] = value;
    return BVHGPU_OK;
}
int bvhgpu_get_tuning(const bvhgpu_ctx* ctx, int knob, int* value) {
    if (!ctx || !value || knob < 0 || knob >= BVHGPU_TUNE_COUNT) return BVHGPU_INVALID_ARG;
    *value = ctx->tune[knob];
    return BVHGPU_OK;
}

#ifdef BVH_LEVEL_PROFILE
void bvhgpu_debug_level_prof(unsigned long long* out, size_t n) { bvhgpu::debug_level_prof(out, n); }
#endif
#ifdef BVH_WIDE_PROFILE
void bvhgpu_debug_wide_prof(unsigned long long* out, size_t n) { bvhgpu::debug_wide_prof(out, n); }
void bvhgpu_debug_wide_util(unsigned long long* out, size_t n) { bvhgpu::debug_wide_util(out, n); }
#endif
#ifdef BVH_SMALL_PROFILE
void bvhgpu_debug_small_prof(unsigned long long* out, size_t n) { bvhgpu::debug_small_prof(out, n); }
#endif
#ifdef BVH_PROFILE_MID
void bvhgpu_debug_mid_prof(unsigned long long* out, int reset) { bvhgpu::debug_mid_prof(out, reset != 0); }
#endif

}  // extern "C"
