// flatten_node.hpp — what the flatten writes for ONE tree node (flatten.hip has the derivation and the kernel over all nodes): shared with the
// builder's wave tier, which flattens the subtrees it has just built itself (build.hip k_small, BVHGPU_TUNE_FLATTEN_INLINE).
#pragma once
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

// PARTS: FLAT = the FlatNode array (reference layout); TRAV = the folded binary array and the binary walk's LDS slot table; WIDE = the wide
// nodes (+ an f64 tree's guide nodes) and their LDS slot table.  All three in one pass, or — BVHGPU_TUNE_FLATTEN_LAZY — the flatten
// behind a build writes what the wide walk reads and the rest follows when something asks for those arrays (ensure_flat_arrays).
constexpr int FLATTEN_FLAT = 1, FLATTEN_WIDE = 2, FLATTEN_TRAV = 4;
template <typename T> struct FlattenArgs {
    const typename Traits<T>::Node* nodes;
    const uint32_t* node_start;
    const uint32_t* node_count;
    const T* aabbs;
    const uint16_t* node_slot;
    uint32_t* slot_entry;
    typename Traits<T>::Flat* flat;
    TravNode<T>* trav;
    WideNode<T>* wide;        // NULL: no wide nodes for this tree
    uint32_t* wslot_node;
    WideNode<float>* guide;   // f64 trees with wide nodes
    float* guide_info;
    uint32_t n_nodes, n_shapes;
};

// everything tree node i (record `nd`) contributes to the arrays PARTS names.  The arrays arrive as __restrict__ PARAMETERS (the no-alias facts
// survive inlining as scoped metadata; as locals or struct members they do not, and the loads of the FLAT part then wait behind the wide node's
// stores: k_flatten 1.63 -> 1.95 ms at 12 M shapes)
template <typename T, int PARTS>
__device__ __forceinline__ void flatten_node_impl(const typename Traits<T>::Node* __restrict__ nodes, const uint32_t* __restrict__ node_start,
                                                  const uint32_t* __restrict__ node_count, const T* __restrict__ aabbs,
                                                  const uint16_t* __restrict__ node_slot, uint32_t* __restrict__ slot_entry,
                                                  typename Traits<T>::Flat* __restrict__ flat, TravNode<T>* __restrict__ trav,
                                                  WideNode<T>* __restrict__ wide, uint32_t* __restrict__ wslot_node, WideNode<float>* __restrict__ guide,
                                                  float* __restrict__ guide_info, const uint32_t n_nodes, const uint32_t n_shapes, const uint32_t i,
                                                  const typename Traits<T>::Node& nd) {
    using Tr = Traits<T>;
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
            typename Tr::Flat fe = {};
            for (int k = 0; k < 3; k++) { fe.min[k] = Tr::inf(); fe.max[k] = -Tr::inf(); }
            fe.entry = NONE; fe.exit = 1; fe.shape = nd.shape;
            flat[0] = fe;
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
        typename Tr::Flat fe = {};
#pragma unroll
        for (int k = 0; k < 3; k++) { fe.min[k] = mn[k]; fe.max[k] = mx[k]; }
        fe.entry = nav + 1;
        fe.exit = nav + 3 * kcnt - 1;
        fe.shape = NONE;
        flat[nav] = fe;
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

template <typename T, int PARTS>
__device__ __forceinline__ void flatten_node(const FlattenArgs<T>& f, const uint32_t i, const typename Traits<T>::Node& nd) {
    flatten_node_impl<T, PARTS>(f.nodes, f.node_start, f.node_count, f.aabbs, f.node_slot, f.slot_entry, f.flat, f.trav, f.wide, f.wslot_node, f.guide,
                                f.guide_info, f.n_nodes, f.n_shapes, i, nd);
}

// (flatten.hip) the arrays of a tree as the flatten's kernels take them
template <typename T> FlattenArgs<T> flatten_args(bvhgpu_tree* t, bool with_wide, bool with_guide);

}  // namespace bvhgpu
