// flatten.hip — Bvh::flatten (src/flat_bvh.rs:60-143, 240-251, 312-319) as a closed-form scatter.
//
// The reference emits the flat array by a serial pre-order recursion with Vec::push.  Because the
// builder keeps, for every tree node i, the first sorted position of its index slice (= number of
// leaves with a smaller pre-order index, L_i) and its shape count k_i, every entry's position is
// known without recursion (verified in tests/test_oracle_golden.py):
//     nav(i)  = i - 1 + L_i                      (navigator entry of node i >= 1, flat_bvh.rs:60-89)
//     entry   = nav + 1, exit = nav + 3*k_i - 1  (index after the subtree)
//     leaf i additionally owns flat[nav+1] = {Aabb::empty(), u32::MAX, nav+2, shape}  (:129-141)
// One thread per tree node; the same thread writes the engine's traversal entry trav[i-1]
// (folded layout, common.hpp) so both arrays come out of one pass over the nodes.
#include <algorithm>

#include "flatten_node.hpp"

namespace bvhgpu {

// One thread per tree node.  inline_parts: what the builder's wave tier has already written for the nodes of its subtrees (every node of at most
// SMALL_MAX shapes: build.hip k_small) — those threads leave after one 4-byte load, or write only what is still missing (TRAV).
template <typename T, int PARTS>
__global__ __launch_bounds__(256) void k_flatten(const typename Traits<T>::Node* __restrict__ nodes,
                                                 const uint32_t* __restrict__ node_start,
                                                 const uint32_t* __restrict__ node_count, const T* __restrict__ aabbs,
                                                 const uint16_t* __restrict__ node_slot, uint32_t* __restrict__ slot_entry,
                                                 typename Traits<T>::Flat* __restrict__ flat, TravNode<T>* __restrict__ trav,
                                                 WideNode<T>* __restrict__ wide, uint32_t* __restrict__ wslot_node,
                                                 uint32_t n_nodes, uint32_t n_shapes, uint32_t* __restrict__ pub_ctr,
                                                 uint32_t* __restrict__ pub_host, uint32_t pub_words, uint32_t* __restrict__ bstat,
                                                 uint32_t flags_idx, uint32_t level_idx, WideNode<float>* __restrict__ guide,
                                                 float* __restrict__ guide_info, uint32_t inline_parts) {
    // build + flatten in one enqueue: this launch is the last of the chain, so its first workgroup also stores the builder's
    // counters in the tree's pinned host page and zeroes them for the next build (nothing in this kernel reads them) —
    // a launch of its own for that cost 4.4 µs of the step
    if (pub_ctr && blockIdx.x == 0) {
        if (bstat) {   // the same facts for a broadcast header composed on the device (comm.hip): flags, unfinished level queue
            if (threadIdx.x == 0) bstat[0] = pub_ctr[flags_idx] | ((level_idx != flags_idx && pub_ctr[level_idx] != 0u) ? BSTAT_UNFINISHED : 0u);
            __syncthreads();   // (read before the counters are zeroed below)
        }
        for (uint32_t k = threadIdx.x; k < pub_words; k += blockDim.x) { pub_host[k] = pub_ctr[k]; pub_ctr[k] = 0; }
        __threadfence_system();
    }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    if (inline_parts != 0u && node_count[i] <= (uint32_t)SMALL_MAX) {   // (inline_parts: uniform)
        // The wide walk's LDS slot table stays this kernel's job even for those nodes: k_prep clears it with every build, and a build that failed
        // (NaN input: no wave tier ran) must leave the walk enqueued behind it a table that matches the — stale but consistent — arrays it will read;
        // a resident node without its table entry is garbage in LDS (a page fault in the walk, found by tests/test_gpu_host.py)
        if ((PARTS & FLATTEN_WIDE) && wide) {
            const uint32_t h = node_slot[i];
            if (h >= 1u && h < 2048u && ((31 - __clz((int)h)) & 1) == 0) {
                const typename Traits<T>::Node nd = nodes[i];
                if (nd.shape == NONE && nd.l < n_nodes && nd.r < n_nodes) {
                    const int level = 31 - __clz((int)h);
                    wslot_node[wide_level_base(level >> 1) + (h - (1u << level))] = i;
                }
            }
        }
        constexpr int REST = PARTS & FLATTEN_TRAV;   // (the wave tier writes FLAT and WIDE, never TRAV: that part reads a node outside the subtree)
        if (REST != 0 && (PARTS & ~(int)inline_parts) == REST) {
            flatten_node_impl<T, REST>(nodes, node_start, node_count, aabbs, node_slot, slot_entry, flat, trav, wide, wslot_node, guide, guide_info, n_nodes,
                                       n_shapes, i, nodes[i]);
            return;
        }
        if ((PARTS & ~(int)inline_parts) == 0) return;
    }
    flatten_node_impl<T, PARTS>(nodes, node_start, node_count, aabbs, node_slot, slot_entry, flat, trav, wide, wslot_node, guide, guide_info, n_nodes, n_shapes,
                                i, nodes[i]);
}

// ------------------------------------------------------------------------------------------------
// Wide nodes (common.hpp WideNode) from the folded traversal array: tree node b (entry b-1; the root has no entry) is
// inner; its left child is entry b, its right child is the entry the left child exits to; the same step down gives the
// grandchildren.  Works for trees built here and for imported scenes alike (the array is all it reads).  Threads
// below WIDE_SLOTS also fill the 4-ary heap slot table of the LDS-resident top of the wide walk: slot q at wide level k
// is the tree node with binary heap number 4^k + (q - base_k), whose traversal entry the binary slot table holds.
// ------------------------------------------------------------------------------------------------
template <typename T> struct WideSlot { T mn[3], mx[3]; uint32_t ref; };

template <typename T> __device__ __forceinline__ void wide_slot_from_entry(const TravNode<T>* __restrict__ trav, uint32_t e,
                                                                           WideSlot<T>& s) {
    const TravNode<T> g = trav[e];
#pragma unroll
    for (int k = 0; k < 3; k++) { s.mn[k] = g.mn[k]; s.mx[k] = g.mx[k]; }
    s.ref = trav_is_leaf(g.shape) ? g.shape : (WIDE_INNER | (e + 1u));   // entry e belongs to tree node e + 1
}
template <typename T> __device__ __forceinline__ void wide_slot_absent(WideSlot<T>& s) {
    const T nan = __builtin_nan("");
#pragma unroll
    for (int k = 0; k < 3; k++) { s.mn[k] = nan; s.mx[k] = nan; }
    s.ref = NONE;
}

template <typename T>
__global__ __launch_bounds__(256) void k_wide(const TravNode<T>* __restrict__ trav, uint32_t n_nodes, uint32_t n_trav,
                                              const uint32_t* __restrict__ slot_entry, uint32_t n_bin_slots,
                                              WideNode<T>* __restrict__ wide, uint32_t* __restrict__ wslot_node,
                                              WideNode<float>* __restrict__ guide, float* __restrict__ guide_info) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < WIDE_SLOTS) {   // slot table of the resident top
        int k = 0;
        while (k < 5 && wide_level_base(k + 1) <= b) k++;
        const uint32_t h = (1u << (2 * k)) + (b - wide_level_base(k));   // binary heap number (root 1)
        uint32_t node = NONE;
        if (h == 1u) node = 0u;
        else if (h < n_bin_slots) {
            const uint32_t e = slot_entry[h];
            if (e != NONE && e < n_trav && !trav_is_leaf(trav[e].shape)) node = e + 1u;
        }
        wslot_node[b] = node;
    }
    if (b >= n_nodes) return;
    if (b > 0 && trav_is_leaf(trav[b - 1].shape)) return;   // leaves have no wide node
    WideSlot<T> s[4];
    // every index read from the array is range-checked: the optimistic flatten of build_flat may run over an unfinished tree
    // (garbage in, garbage out — the host flattens again — but never out of range)
    uint32_t child = b < n_trav ? b : n_trav - 1u;   // entry of the left child
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const TravNode<T> c = trav[child];
        if (trav_is_leaf(c.shape) || child + 1u >= n_trav) {
#pragma unroll
            for (int k = 0; k < 3; k++) { s[2 * side].mn[k] = c.mn[k]; s[2 * side].mx[k] = c.mx[k]; }
            s[2 * side].ref = trav_is_leaf(c.shape) ? c.shape : NONE;
            wide_slot_absent<T>(s[2 * side + 1]);
        } else {
            wide_slot_from_entry<T>(trav, child + 1u, s[2 * side]);
            const uint32_t g1 = trav[child + 1u].exit;
            if (g1 < n_trav) wide_slot_from_entry<T>(trav, g1, s[2 * side + 1]); else wide_slot_absent<T>(s[2 * side + 1]);
        }
        child = c.exit < n_trav ? c.exit : n_trav - 1u;   // the left child exits to the right child
    }
    WideNode<T> w;
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int k = 0; k < 3; k++) { w.mn[k][c] = s[c].mn[k]; w.mx[k][c] = s[c].mx[k]; }
        w.ref[c] = s[c].ref;
    }
    for (int k = 0; k < (int)(sizeof(w._pad) / 4); k++) w._pad[k] = 0;
    wide[b] = w;
    if (sizeof(T) == 8 && guide && n_trav >= 2u) {   // the root's children: entry 0 and the entry it exits to
        const TravNode<T> c0 = trav[0];
        const TravNode<T> c1 = trav[c0.exit < n_trav ? c0.exit : n_trav - 1u];
        const double S = guide_scene_extent<T>(c0.mn, c0.mx, c1.mn, c1.mx);
        guide[b] = guide_node(w, GUIDE_GROW * S);
        if (b == 0 && guide_info) guide_info[0] = (float)S;
    }
}

template <typename T> void wide_from_trav(bvhgpu_tree* t) {
    t->has_wide = false; t->has_guide = false;
    if (t->n < 2 || t->n >= WIDE_MAX_SHAPES || t->unfolded || !t->slot_entry.p) return;
    const uint32_t nn = (uint32_t)(t->n_trav + 1);   // tree nodes = entries + the root
    t->wide.reserve((size_t)nn * sizeof(WideNode<T>));
    t->wslot_node.reserve(WIDE_SLOTS * 4);
    const bool with_guide = sizeof(T) == 8;
    if (with_guide) { t->wide_guide.reserve((size_t)nn * sizeof(WideNode<float>)); t->guide_info.reserve(16); }
    hipLaunchKernelGGL(k_wide<T>, dim3((std::max(nn, WIDE_SLOTS) + 255) / 256), dim3(256), 0, t->ctx->stream, t->trav.as<TravNode<T>>(),
                       nn, (uint32_t)t->n_trav, t->slot_entry.as<uint32_t>(), (uint32_t)TopCfg<T>::SLOTS, t->wide.as<WideNode<T>>(),
                       t->wslot_node.as<uint32_t>(), with_guide ? t->wide_guide.as<WideNode<float>>() : nullptr,
                       with_guide ? t->guide_info.as<float>() : nullptr);
    BVH_HIP(hipGetLastError());
    t->has_wide = true;
    t->has_guide = with_guide;
}
template void wide_from_trav<float>(bvhgpu_tree*);
template void wide_from_trav<double>(bvhgpu_tree*);

template <typename T> FlattenArgs<T> flatten_args(bvhgpu_tree* t, bool with_wide, bool with_guide) {
    using Tr = Traits<T>;
    FlattenArgs<T> f;
    f.nodes = t->nodes.as<typename Tr::Node>(); f.node_start = t->node_start.as<uint32_t>(); f.node_count = t->node_count.as<uint32_t>();
    f.aabbs = t->aabbs.as<T>(); f.node_slot = t->node_slot.as<uint16_t>(); f.slot_entry = t->slot_entry.as<uint32_t>();
    f.flat = t->flat.as<typename Tr::Flat>(); f.trav = t->trav.as<TravNode<T>>();
    f.wide = with_wide ? t->wide.as<WideNode<T>>() : nullptr; f.wslot_node = t->wslot_node.as<uint32_t>();
    f.guide = with_guide ? t->wide_guide.as<WideNode<float>>() : nullptr; f.guide_info = with_guide ? t->guide_info.as<float>() : nullptr;
    f.n_nodes = (uint32_t)t->n_nodes; f.n_shapes = (uint32_t)t->n;
    return f;
}
template FlattenArgs<float> flatten_args<float>(bvhgpu_tree*, bool, bool);
template FlattenArgs<double> flatten_args<double>(bvhgpu_tree*, bool, bool);

// Which parts the flatten behind a build of this tree writes (BVHGPU_TUNE_FLATTEN_LAZY), and their arrays reserved: what flatten_tree will
// launch — asked ahead of the build's launches by build_enqueue, whose wave tier writes FLAT and WIDE for the subtrees it builds.
template <typename T> FlattenPlan flatten_plan(bvhgpu_tree* t, bool wide_only) {
    using Tr = Traits<T>;
    FlattenPlan p;
    p.with_wide = t->n >= 2 && t->n < WIDE_MAX_SHAPES && t->wslot_node.p != nullptr;
    p.with_guide = p.with_wide && sizeof(T) == 8;
    const uint32_t nn = (uint32_t)t->n_nodes;
    // a fresh wide array is all absent slots (NaN boxes, NONE references): whatever the optimistic flatten of an unfinished or failed build leaves
    // unwritten, the walk enqueued behind it reads nothing it could follow out of range
    if (p.with_wide && t->wide.reserve((size_t)nn * sizeof(WideNode<T>))) BVH_HIP(hipMemsetAsync(t->wide.p, 0xFF, t->wide.cap, t->ctx->stream));
    if (p.with_guide) { t->wide_guide.reserve((size_t)nn * sizeof(WideNode<float>)); t->guide_info.reserve(16); }
    if (wide_only && p.with_wide && t->ctx->tune[BVHGPU_TUNE_FLATTEN_LAZY] == 3) p.parts = FLATTEN_FLAT | FLATTEN_WIDE;
    else if (wide_only && p.with_wide) p.parts = FLATTEN_WIDE;   // (a tree without wide nodes is walked by the binary kernels: nothing to postpone)
    else p.parts = FLATTEN_FLAT | FLATTEN_TRAV | FLATTEN_WIDE;
    if (p.parts & FLATTEN_FLAT) t->flat.reserve(t->n_flat * sizeof(typename Tr::Flat));
    if (p.parts & FLATTEN_TRAV) t->trav.reserve(t->n_trav * sizeof(TravNode<T>));
    return p;
}
template FlattenPlan flatten_plan<float>(bvhgpu_tree*, bool);
template FlattenPlan flatten_plan<double>(bvhgpu_tree*, bool);

template <typename T, int PARTS> static void launch_flatten(bvhgpu_tree* t, bool with_wide, bool with_guide, uint32_t* pub_ctr, uint32_t* pub_host,
                                                             uint32_t pub_words, uint32_t* bstat, uint32_t flags_idx, uint32_t level_idx,
                                                             hipStream_t st = nullptr, uint32_t inline_parts = 0) {
    const uint32_t nn = (uint32_t)t->n_nodes;
    const FlattenArgs<T> f = flatten_args<T>(t, with_wide, with_guide);
    hipLaunchKernelGGL((k_flatten<T, PARTS>), dim3((nn + 255) / 256), dim3(256), 0, st ? st : t->ctx->stream, f.nodes, f.node_start, f.node_count, f.aabbs,
                       f.node_slot, f.slot_entry, f.flat, f.trav, f.wide, f.wslot_node, nn, (uint32_t)t->n, pub_ctr, pub_host, pub_words, bstat, flags_idx,
                       level_idx, f.guide, f.guide_info, inline_parts);
    BVH_HIP(hipGetLastError());
}

void join_flat(bvhgpu_tree* t) {
    if (!t->flat_beside) return;
    t->flat_beside = false;
    BVH_HIP(hipStreamWaitEvent(t->ctx->stream, t->ev_flat, 0));
}

// BVHGPU_TUNE_FLATTEN_LAZY = 2: part 1 of the flatten at once, but on the ctx's side stream — the walk that follows on the main stream reads
// none of what it writes (flat / trav / the binary slot table), both only read the BvhNode array, and the 20 MB it streams out fit beside
// a walk that is bound by instruction issue
template <typename T> static void flat_beside(bvhgpu_tree* t) {
    bvhgpu_ctx* ctx = t->ctx;
    if (!ctx->side) BVH_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    if (!t->ev_flat0) BVH_HIP(hipEventCreateWithFlags(&t->ev_flat0, hipEventDisableTiming));
    if (!t->ev_flat) BVH_HIP(hipEventCreateWithFlags(&t->ev_flat, hipEventDisableTiming));
    t->flat.reserve(t->n_flat * sizeof(typename Traits<T>::Flat));
    t->trav.reserve(t->n_trav * sizeof(TravNode<T>));
    BVH_HIP(hipEventRecord(t->ev_flat0, ctx->stream));
    BVH_HIP(hipStreamWaitEvent(ctx->side, t->ev_flat0, 0));
    launch_flatten<T, FLATTEN_FLAT | FLATTEN_TRAV>(t, false, false, nullptr, nullptr, 0, nullptr, 0, 0, ctx->side);
    BVH_HIP(hipEventRecord(t->ev_flat, ctx->side));
    t->lazy_flat = false;
    t->flat_beside = true;
}

template <typename T> void flatten_tree(bvhgpu_tree* t, uint32_t* pub_ctr, uint32_t* pub_host, uint32_t pub_words, uint32_t* bstat,
                                        uint32_t flags_idx, uint32_t level_idx, bool wide_only, uint32_t inline_parts) {
    join_flat(t);
    t->lazy_flat = false;
    if (t->n == 0) { t->flattened = true; return; }
    // the wide nodes come out of the same pass (their LDS slot table was cleared by the build's first kernel)
    const FlattenPlan p = flatten_plan<T>(t, wide_only);
    const bool with_wide = p.with_wide, with_guide = p.with_guide;
    if (p.parts == (FLATTEN_FLAT | FLATTEN_WIDE)) {
        // the reference's FlatNode array at once (what Bvh::flatten returns), the engine's own folded binary array on first use
        launch_flatten<T, FLATTEN_FLAT | FLATTEN_WIDE>(t, true, with_guide, pub_ctr, pub_host, pub_words, bstat, flags_idx, level_idx, nullptr, inline_parts);
        t->lazy_flat = true; t->lazy_parts = FLATTEN_TRAV;
    } else if (p.parts == FLATTEN_WIDE) {
        launch_flatten<T, FLATTEN_WIDE>(t, true, with_guide, pub_ctr, pub_host, pub_words, bstat, flags_idx, level_idx, nullptr, inline_parts);
        t->lazy_flat = true; t->lazy_parts = FLATTEN_FLAT | FLATTEN_TRAV;
        if (t->ctx->tune[BVHGPU_TUNE_FLATTEN_LAZY] == 2) flat_beside<T>(t);
    } else {
        launch_flatten<T, FLATTEN_FLAT | FLATTEN_TRAV | FLATTEN_WIDE>(t, with_wide, with_guide, pub_ctr, pub_host, pub_words, bstat, flags_idx, level_idx, nullptr, inline_parts);
    }
    t->has_wide = with_wide;
    t->has_guide = with_guide;
    t->flattened = true;
}

// part 1 of a lazy flatten, on the tree's stream (behind the build and the wide-only pass if they are still in flight: on an unfinished
// tree build_finalize flattens again, completely).  The slot table k_prep cleared is filled here.
void ensure_flat_arrays(bvhgpu_tree* t) {
    join_flat(t);
    if (!t->lazy_flat) return;
    t->lazy_flat = false;
    const bool trav_only = t->lazy_parts == FLATTEN_TRAV;   // (BVHGPU_TUNE_FLATTEN_LAZY = 3: the FlatNode array is there already)
    if (t->dtype == BVHGPU_F32) {
        t->flat.reserve(t->n_flat * sizeof(Traits<float>::Flat));
        t->trav.reserve(t->n_trav * sizeof(TravNode<float>));
        if (trav_only) launch_flatten<float, FLATTEN_TRAV>(t, false, false, nullptr, nullptr, 0, nullptr, 0, 0);
        else launch_flatten<float, FLATTEN_FLAT | FLATTEN_TRAV>(t, false, false, nullptr, nullptr, 0, nullptr, 0, 0);
    } else {
        t->flat.reserve(t->n_flat * sizeof(Traits<double>::Flat));
        t->trav.reserve(t->n_trav * sizeof(TravNode<double>));
        if (trav_only) launch_flatten<double, FLATTEN_TRAV>(t, false, false, nullptr, nullptr, 0, nullptr, 0, 0);
        else launch_flatten<double, FLATTEN_FLAT | FLATTEN_TRAV>(t, false, false, nullptr, nullptr, 0, nullptr, 0, 0);
    }
}

template void flatten_tree<float>(bvhgpu_tree*, uint32_t*, uint32_t*, uint32_t, uint32_t*, uint32_t, uint32_t, bool, uint32_t);
template void flatten_tree<double>(bvhgpu_tree*, uint32_t*, uint32_t*, uint32_t, uint32_t*, uint32_t, uint32_t, bool, uint32_t);

}  // namespace bvhgpu
