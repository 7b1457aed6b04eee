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

#include "engine.hpp"

namespace bvhgpu {

template <typename T> __device__ __forceinline__ void write_trav(TravNode<T>* tn, const T* mn, const T* mx, uint32_t exit_,
                                                                  uint32_t shape);
template <> __device__ __forceinline__ void write_trav<float>(TravNode<float>* tn, const float* mn, const float* mx,
                                                              uint32_t exit_, uint32_t shape) {
    float4* p = reinterpret_cast<float4*>(tn);
    p[0] = make_float4(mn[0], mn[1], mn[2], __uint_as_float(exit_));
    p[1] = make_float4(mx[0], mx[1], mx[2], __uint_as_float(shape));
}
template <> __device__ __forceinline__ void write_trav<double>(TravNode<double>* tn, const double* mn, const double* mx,
                                                               uint32_t exit_, uint32_t shape) {
    double2* p = reinterpret_cast<double2*>(tn);
    p[0] = make_double2(mn[0], mn[1]);
    p[1] = make_double2(mn[2], mx[0]);
    p[2] = make_double2(mx[1], mx[2]);
    unsigned long long es = (unsigned long long)exit_ | ((unsigned long long)shape << 32);
    p[3] = make_double2(__longlong_as_double((long long)es), 0.0);
}

// f64 wide node → its f32 guide node (common.hpp "guide boxes"); absent slots keep their NaN boxes
__device__ __forceinline__ WideNode<float> guide_node(const WideNode<double>& w, double delta) {
    WideNode<float> g;
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int c = 0; c < 4; c++) { g.mn[k][c] = f32_below(w.mn[k][c] - delta); g.mx[k][c] = f32_above(w.mx[k][c] + delta); }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) g.ref[c] = w.ref[c];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(g._pad) / 4); k++) g._pad[k] = 0;
    return g;
}
__device__ __forceinline__ WideNode<float> guide_node(const WideNode<float>& w, double) { return w; }   // (never used: f32 trees have no guide)
// S of a tree from the two child boxes of its root (their union is the scene)
template <typename T> __device__ __forceinline__ double guide_scene_extent(const T* a, const T* b, const T* c, const T* d) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) s = fmax(fmax(s, fmax(fabs((double)a[k]), fabs((double)b[k]))), fmax(fabs((double)c[k]), fabs((double)d[k])));
    return s;
}

// The wide node (common.hpp WideNode) of inner tree node i, straight from the BvhNode array: slots 0,1 = the left child's
// children (or the left child itself when it is a leaf), slots 2,3 likewise on the right.  A child's box is its parent's
// child_l_aabb / child_r_aabb; a leaf's is bit-identical to its shape's AABB (join(empty, aabb) == aabb).
template <typename T>
__device__ __forceinline__ void flatten_wide_node(const typename Traits<T>::Node* __restrict__ nodes, const typename Traits<T>::Node& nd,
                                                  uint32_t i, const uint16_t* __restrict__ node_slot, WideNode<T>* __restrict__ wide,
                                                  uint32_t* __restrict__ wslot_node, uint32_t n_nodes, uint32_t n_shapes,
                                                  WideNode<float>* __restrict__ guide) {
    const T nan = __builtin_nan("");
    const typename Traits<T>::Node cl = nodes[nd.l], cr = nodes[nd.r];
    // references of the four grandchildren: a leaf by its shape, an inner node by its index
    uint32_t gidx[4] = {cl.l, cl.r, cr.l, cr.r};
    uint32_t gshape[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const bool has = (c < 2 ? cl.shape : cr.shape) == NONE && gidx[c] < n_nodes;
        gshape[c] = has ? nodes[gidx[c]].shape : NONE;
    }
    WideNode<T> w;
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const typename Traits<T>::Node& c = side ? cr : cl;
        const T* cmn = side ? nd.r_min : nd.l_min;
        const T* cmx = side ? nd.r_max : nd.l_max;
        if (c.shape != NONE) {   // the child is a leaf
#pragma unroll
            for (int k = 0; k < 3; k++) { w.mn[k][2 * side] = cmn[k]; w.mx[k][2 * side] = cmx[k]; w.mn[k][2 * side + 1] = nan; w.mx[k][2 * side + 1] = nan; }
            w.ref[2 * side] = c.shape < n_shapes ? c.shape : NONE;
            w.ref[2 * side + 1] = NONE;
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                w.mn[k][2 * side] = c.l_min[k]; w.mx[k][2 * side] = c.l_max[k];
                w.mn[k][2 * side + 1] = c.r_min[k]; w.mx[k][2 * side + 1] = c.r_max[k];
            }
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const uint32_t gi = gidx[2 * side + g], gs = gshape[2 * side + g];
                w.ref[2 * side + g] = gi >= n_nodes ? NONE : (gs != NONE ? (gs < n_shapes ? gs : NONE) : (WIDE_INNER | gi));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < (int)(sizeof(w._pad) / 4); k++) w._pad[k] = 0;
    wide[i] = w;
    if (sizeof(T) == 8 && guide) {
        const typename Traits<T>::Node& r0 = nodes[0];
        guide[i] = guide_node(w, GUIDE_GROW * guide_scene_extent<T>(r0.l_min, r0.l_max, r0.r_min, r0.r_max));
    }
    // LDS slot table of the wide walk: tree levels 0, 2, .., 10 in 4-ary heap order (binary heap number h: root 1)
    const uint32_t h = node_slot[i];
    if (h >= 1u && h < 2048u) {
        const int level = 31 - __clz((int)h);
        if ((level & 1) == 0) wslot_node[wide_level_base(level >> 1) + (h - (1u << level))] = i;
    }
}

// PARTS: bit 0 = the FlatNode array (reference layout), the folded binary array and the binary walk's LDS slot table; bit 1 = the wide
// nodes (+ an f64 tree's guide nodes) and their LDS slot table.  3 = everything in one pass; with BVHGPU_TUNE_FLATTEN_LAZY the flatten
// behind a build runs part 2 — what the wide walk reads — and part 1 follows when something asks for those arrays (ensure_flat_arrays).
constexpr int FLATTEN_FLAT = 1, FLATTEN_WIDE = 2, FLATTEN_TRAV = 4;   // FLAT: the reference-layout FlatNode array; TRAV: the folded binary array + its LDS slot table
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
                                                 float* __restrict__ guide_info) {
    using Tr = Traits<T>;
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
    const typename Tr::Node nd = nodes[i];
    // wide nodes: the walk only ever enters nodes an even number of levels below the root (it steps from a node to its
    // grandchildren, items start 2 or 4 levels down), so odd levels get none — as far as the level is known: heap numbers
    // saturate 16 levels down, below that every inner node gets one
    if ((PARTS & FLATTEN_WIDE) && wide && nd.shape == NONE && nd.l < n_nodes && nd.r < n_nodes) {
        const uint32_t h = node_slot[i];
        const bool odd_level = h != SLOT_NONE && h >= 1u && (((31 - __clz((int)h)) & 1) != 0);
        if (!odd_level) flatten_wide_node<T>(nodes, nd, i, node_slot, wide, wslot_node, n_nodes, n_shapes, guide);
        if (guide_info && i == 0) guide_info[0] = (float)guide_scene_extent<T>(nd.l_min, nd.l_max, nd.r_min, nd.r_max);   // (read by the ray conversion of the guide walk)
    }
    if (!(PARTS & (FLATTEN_FLAT | FLATTEN_TRAV))) return;
    if (n_nodes == 1) {
        // single-shape tree: the root is a leaf and emits one leaf entry (flat_bvh.rs:129-141); its
        // traversal entry tests the shape's own AABB (flat_bvh.rs:411-418)
        if (PARTS & FLATTEN_FLAT) {
            typename Tr::Flat f = {};
            for (int k = 0; k < 3; k++) { f.min[k] = Tr::inf(); f.max[k] = -Tr::inf(); }
            f.entry = NONE; f.exit = 1; f.shape = nd.shape;
            flat[0] = f;
        }
        if (PARTS & FLATTEN_TRAV) {
            const T* sb = aabbs + 6 * (size_t)nd.shape;
            write_trav<T>(&trav[0], sb, sb + 3, 1u, nd.shape);
        }
        return;
    }
    if (i == 0) return;  // the root emits nothing itself (flat_bvh.rs:104-127)
    // build_flat launches this kernel optimistically, before the host has seen that the builder's queues are drained.
    // On an unfinished (very unbalanced) tree some nodes are not written yet: nothing read from such a node may turn
    // into an out-of-range access; the host flattens again once the build is complete.
    const uint32_t n_flat = 3u * n_shapes - 2u;
    if (nd.parent >= n_nodes || (nd.shape != NONE && nd.shape >= n_shapes)) return;
    if ((unsigned long long)(i - 1) + node_start[i] + 1ull >= n_flat || node_count[i] > n_shapes) return;
    const typename Tr::Node pn = nodes[nd.parent];
    const bool is_left = pn.l == i;
    T mn[3], mx[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        mn[k] = is_left ? pn.l_min[k] : pn.r_min[k];
        mx[k] = is_left ? pn.l_max[k] : pn.r_max[k];
    }
    const uint32_t L = node_start[i], kcnt = node_count[i];
    // the builder numbered the nodes heap-style (root 1, children 2h / 2h+1): the first TopCfg<T>::SLOTS of
    // them are the top of the tree that traversal keeps in LDS; slot h holds traversal entry i-1
    const bool leaf = nd.shape != NONE;
    if (PARTS & FLATTEN_FLAT) {
        const uint32_t nav = i - 1 + L;
        typename Tr::Flat f = {};
#pragma unroll
        for (int k = 0; k < 3; k++) { f.min[k] = mn[k]; f.max[k] = mx[k]; }
        f.entry = nav + 1;
        f.exit = nav + 3 * kcnt - 1;
        f.shape = NONE;
        flat[nav] = f;
        if (leaf) {
            typename Tr::Flat lf = {};
#pragma unroll
            for (int k = 0; k < 3; k++) { lf.min[k] = Tr::inf(); lf.max[k] = -Tr::inf(); }
            lf.entry = NONE; lf.exit = nav + 2; lf.shape = nd.shape;
            flat[nav + 1] = lf;
        }
    }
    if (PARTS & FLATTEN_TRAV) {
        const uint32_t myslot = node_slot[i];
        if (myslot < TopCfg<T>::SLOTS) slot_entry[myslot] = i - 1;
        if (leaf) {
            // folded leaf: one test against the shape's own AABB.  For a tree built here it is
            // bit-identical to the navigator box (join(empty, aabb) == aabb), so nav-then-leaf of
            // flat_bvh.rs:411-427 collapses to a single slab test with the same outcome.
            const T* sb = aabbs + 6 * (size_t)nd.shape;
            write_trav<T>(&trav[i - 1], sb, sb + 3, i, nd.shape);
        } else {
            const uint32_t ex = (i - 1) + (2 * kcnt - 1);                       // first entry after the subtree
            const uint32_t exs = ex + 1 < n_nodes ? (uint32_t)node_slot[ex + 1] : SLOT_NONE;  // entry ex belongs to tree node ex+1
            write_trav<T>(&trav[i - 1], mn, mx, ex, TRAV_INNER | exs);
        }
    }
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

template <typename T, int PARTS> static void launch_flatten(bvhgpu_tree* t, bool with_wide, bool with_guide, uint32_t* pub_ctr, uint32_t* pub_host,
                                                             uint32_t pub_words, uint32_t* bstat, uint32_t flags_idx, uint32_t level_idx,
                                                             hipStream_t st = nullptr) {
    using Tr = Traits<T>;
    const uint32_t nn = (uint32_t)t->n_nodes;
    hipLaunchKernelGGL((k_flatten<T, PARTS>), dim3((nn + 255) / 256), dim3(256), 0, st ? st : t->ctx->stream,
                       t->nodes.as<typename Tr::Node>(), t->node_start.as<uint32_t>(), t->node_count.as<uint32_t>(),
                       t->aabbs.as<T>(), t->node_slot.as<uint16_t>(), t->slot_entry.as<uint32_t>(),
                       t->flat.as<typename Tr::Flat>(),
                       t->trav.as<TravNode<T>>(), with_wide ? t->wide.as<WideNode<T>>() : nullptr, t->wslot_node.as<uint32_t>(), nn,
                       (uint32_t)t->n, pub_ctr, pub_host, pub_words, bstat, flags_idx, level_idx,
                       with_guide ? t->wide_guide.as<WideNode<float>>() : nullptr, with_guide ? t->guide_info.as<float>() : nullptr);
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
                                        uint32_t flags_idx, uint32_t level_idx, bool wide_only) {
    using Tr = Traits<T>;
    join_flat(t);
    t->lazy_flat = false;
    if (t->n == 0) { t->flattened = true; return; }
    // the wide nodes come out of the same pass (their LDS slot table was cleared by the build's first kernel)
    const bool with_wide = t->n >= 2 && t->n < WIDE_MAX_SHAPES && t->wslot_node.p != nullptr;
    const uint32_t nn = (uint32_t)t->n_nodes;
    if (with_wide) t->wide.reserve((size_t)nn * sizeof(WideNode<T>));
    const bool with_guide = with_wide && sizeof(T) == 8;
    if (with_guide) { t->wide_guide.reserve((size_t)nn * sizeof(WideNode<float>)); t->guide_info.reserve(16); }
    if (wide_only && with_wide && t->ctx->tune[BVHGPU_TUNE_FLATTEN_LAZY] == 3) {
        // the reference's FlatNode array at once (what Bvh::flatten returns), the engine's own folded binary array on first use
        t->flat.reserve(t->n_flat * sizeof(typename Tr::Flat));
        launch_flatten<T, FLATTEN_FLAT | FLATTEN_WIDE>(t, true, with_guide, pub_ctr, pub_host, pub_words, bstat, flags_idx, level_idx);
        t->lazy_flat = true; t->lazy_parts = FLATTEN_TRAV;
    } else if (wide_only && with_wide) {   // (a tree without wide nodes is walked by the binary kernels: nothing to postpone)
        launch_flatten<T, FLATTEN_WIDE>(t, true, with_guide, pub_ctr, pub_host, pub_words, bstat, flags_idx, level_idx);
        t->lazy_flat = true; t->lazy_parts = FLATTEN_FLAT | FLATTEN_TRAV;
        if (t->ctx->tune[BVHGPU_TUNE_FLATTEN_LAZY] == 2) flat_beside<T>(t);
    } else {
        t->flat.reserve(t->n_flat * sizeof(typename Tr::Flat));
        t->trav.reserve(t->n_trav * sizeof(TravNode<T>));
        launch_flatten<T, FLATTEN_FLAT | FLATTEN_TRAV | FLATTEN_WIDE>(t, with_wide, with_guide, pub_ctr, pub_host, pub_words, bstat, flags_idx, level_idx);
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

template void flatten_tree<float>(bvhgpu_tree*, uint32_t*, uint32_t*, uint32_t, uint32_t*, uint32_t, uint32_t, bool);
template void flatten_tree<double>(bvhgpu_tree*, uint32_t*, uint32_t*, uint32_t, uint32_t*, uint32_t, uint32_t, bool);

}  // namespace bvhgpu
