// refit.hip — the shapes moved, the topology stays: every inner node's child_l_aabb / child_r_aabb becomes the
// exact join of the (new) AABBs of the shapes below that child.
//
// The reference's primitive for this is Bvh::fix_aabbs_ascending (bvh/optimization.rs:355-391): a parent's child
// boxes are re-set to the children's get_node_aabb (bvh_node.rs:616-625: leaf = shapes[i].aabb(), inner =
// child_l_aabb.join(child_r_aabb)), walking up from a changed node.  Applied to every node, bottom-up, that is a
// refit of the whole tree.  (update_shapes, optimization.rs:337-352, removes and re-inserts the changed shapes one
// at a time — a serial pointer chase that also changes the topology; here moved shapes are answered by this refit,
// ~10x cheaper than a build, or by a full rebuild when the topology should follow the motion.)
//
// No bottom-up pointer chase on the GPU: a tree built here keeps, for every node, the range of SORTED POSITIONS of
// its shapes (node_start, node_count: the leaves of a subtree are contiguous in pre-order), so a child box is a
// range join over position-ordered leaf boxes.  min/max are exact, associative, commutative and idempotent, so any
// evaluation order gives the reference's bits (zeros: -0 < +0, as everywhere in the builder).  Three steps:
//   k_refit_leaves  box of sorted position p (and the tree's own copy of the shape AABBs)
//   k_refit_segs    a complete binary tree of joins over the positions (heap layout, 1024-leaf groups per
//                   workgroup reduced in LDS; one more launch per 10 levels)
//   k_refit_nodes   per inner node two range queries (<= 2 log2 N boxes each, L2-resident) → child boxes
#include "engine.hpp"

namespace bvhgpu {

constexpr uint32_t SEG_GROUP = 1024;   // leaves per workgroup of a k_refit_segs pass (10 levels)

template <typename T>
__global__ __launch_bounds__(256) void k_refit_leaves(const T* __restrict__ new_aabbs, const uint32_t* __restrict__ shape_node,
                                                      const uint32_t* __restrict__ node_start, T* __restrict__ tree_aabbs,
                                                      T* __restrict__ seg, uint32_t n, uint32_t n_pad) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_pad) return;
    if (s < n) {
        const uint32_t pos = node_start[shape_node[s]];
        T b[6];
#pragma unroll
        for (int k = 0; k < 6; k++) b[k] = new_aabbs[6 * (size_t)s + k];
#pragma unroll
        for (int k = 0; k < 6; k++) seg[6 * ((size_t)n_pad + pos) + k] = b[k];
        if (tree_aabbs) {   // NULL: the caller's array already IS the tree's copy
#pragma unroll
            for (int k = 0; k < 6; k++) tree_aabbs[6 * (size_t)s + k] = b[k];
        }
    } else {   // padding positions: Aabb::empty (aabb_impl.rs:119-124), the identity of join
#pragma unroll
        for (int k = 0; k < 6; k++) seg[6 * ((size_t)n_pad + s) + k] = k < 3 ? Traits<T>::inf() : -Traits<T>::inf();
    }
}

// One pass: the level with `lv` boxes (heap indices [lv, 2 lv)) is reduced by up to 10 levels; workgroup b owns the
// `group` boxes from lv + b*group and writes every ancestor inside its group: level d above has group>>d boxes at
// heap index (lv >> d) + b*(group>>d) + j.
template <typename T>
__global__ __launch_bounds__(256) void k_refit_segs(T* __restrict__ seg, uint32_t lv, uint32_t group, int levels) {
    __shared__ T s_box[SEG_GROUP * 6];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const T* src = seg + 6 * ((size_t)lv + (size_t)b * group);
    for (uint32_t e = tid; e < group * 6; e += 256) s_box[e] = src[e];
    __syncthreads();
    for (int d = 1; d <= levels; d++) {
        const uint32_t cnt = group >> d;
        T out[(SEG_GROUP / 2 / 256) > 0 ? (SEG_GROUP / 2 / 256) : 1][6];
        int m = 0;
        for (uint32_t j = tid; j < cnt; j += 256, m++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                out[m][k] = join_min(s_box[6 * (2 * j) + k], s_box[6 * (2 * j + 1) + k]);
                out[m][3 + k] = join_max(s_box[6 * (2 * j) + 3 + k], s_box[6 * (2 * j + 1) + 3 + k]);
            }
        }
        __syncthreads();
        T* dst = seg + 6 * ((size_t)(lv >> d) + (size_t)b * cnt);
        m = 0;
        for (uint32_t j = tid; j < cnt; j += 256, m++) {
#pragma unroll
            for (int k = 0; k < 6; k++) { s_box[6 * j + k] = out[m][k]; dst[6 * (size_t)j + k] = out[m][k]; }
        }
        __syncthreads();
    }
}

// join of the position boxes [l, r): the classic bottom-up walk over the heap
template <typename T>
__device__ __forceinline__ void seg_query(const T* __restrict__ seg, uint32_t n_pad, uint32_t l, uint32_t r, T out[6]) {
#pragma unroll
    for (int k = 0; k < 6; k++) out[k] = k < 3 ? Traits<T>::inf() : -Traits<T>::inf();
    l += n_pad; r += n_pad;
    while (l < r) {
        if (l & 1u) {
            const T* p = seg + 6 * (size_t)l;
#pragma unroll
            for (int k = 0; k < 3; k++) { out[k] = join_min(out[k], p[k]); out[3 + k] = join_max(out[3 + k], p[3 + k]); }
            l++;
        }
        if (r & 1u) {
            r--;
            const T* p = seg + 6 * (size_t)r;
#pragma unroll
            for (int k = 0; k < 3; k++) { out[k] = join_min(out[k], p[k]); out[3 + k] = join_max(out[3 + k], p[3 + k]); }
        }
        l >>= 1; r >>= 1;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_refit_nodes(typename Traits<T>::Node* __restrict__ nodes,
                                                     const uint32_t* __restrict__ node_start,
                                                     const uint32_t* __restrict__ node_count, const T* __restrict__ seg,
                                                     uint32_t n_pad, uint32_t n_nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    typename Traits<T>::Node* nd = nodes + i;
    if (nd->shape != NONE) return;   // a leaf stores no box (bvh_node.rs:38-46)
    const uint32_t s = node_start[i], c = node_count[i];
    const uint32_t nl = node_count[nd->l];
    T lb[6], rb[6];
    seg_query<T>(seg, n_pad, s, s + nl, lb);
    seg_query<T>(seg, n_pad, s + nl, s + c, rb);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        nd->l_min[k] = lb[k]; nd->l_max[k] = lb[3 + k];
        nd->r_min[k] = rb[k]; nd->r_max[k] = rb[3 + k];
    }
}

template <typename T> void refit_tree(bvhgpu_tree* t, const T* aabbs_dev) {
    using Tr = Traits<T>;
    const size_t n = t->n;
    if (n == 0) return;
    hipStream_t st = t->ctx->stream;
    if (t->ctx->timing) { BVH_HIP(hipEventRecord(t->ctx->ev[0], st)); }
    if (n == 1) {   // the root is the leaf: only the shape's AABB is stored (and tested by traversal)
        if (aabbs_dev != t->aabbs.as<T>()) BVH_HIP(hipMemcpyAsync(t->aabbs.p, aabbs_dev, 6 * sizeof(T), hipMemcpyDeviceToDevice, st));
        if (t->ctx->timing) { BVH_HIP(hipEventRecord(t->ctx->ev[1], st)); t->ctx->ev_set |= 1u; }
        if (t->flattened) flatten_tree<T>(t, nullptr, nullptr, 0, nullptr, 0, 0, t->ctx->tune[BVHGPU_TUNE_FLATTEN_LAZY] != 0);
        return;
    }
    uint32_t n_pad = 1;
    while (n_pad < n) n_pad <<= 1;
    t->refit_seg.reserve((size_t)2 * n_pad * 6 * sizeof(T));
    T* seg = t->refit_seg.as<T>();
    hipLaunchKernelGGL(k_refit_leaves<T>, dim3((n_pad + 255) / 256), dim3(256), 0, st, aabbs_dev, t->shape_node.as<uint32_t>(),
                       t->node_start.as<uint32_t>(), aabbs_dev == t->aabbs.as<T>() ? nullptr : t->aabbs.as<T>(), seg, (uint32_t)n, n_pad);
    for (uint32_t lv = n_pad; lv > 1;) {
        const uint32_t group = std::min<uint32_t>(SEG_GROUP, lv);
        int levels = 0;
        while ((1u << levels) < group) levels++;
        hipLaunchKernelGGL(k_refit_segs<T>, dim3(lv / group), dim3(256), 0, st, seg, lv, group, levels);
        lv /= group;
    }
    const uint32_t nn = (uint32_t)t->n_nodes;
    hipLaunchKernelGGL(k_refit_nodes<T>, dim3((nn + 255) / 256), dim3(256), 0, st, t->nodes.as<typename Tr::Node>(),
                       t->node_start.as<uint32_t>(), t->node_count.as<uint32_t>(), seg, n_pad, nn);
    BVH_HIP(hipGetLastError());
    if (t->ctx->timing) { BVH_HIP(hipEventRecord(t->ctx->ev[1], st)); t->ctx->ev_set |= 1u; }
    t->exact_only = false;   // every child box is now the exact join of what is below it (also where the build left empty bounds)
    if (t->flattened) flatten_tree<T>(t, nullptr, nullptr, 0, nullptr, 0, 0, t->ctx->tune[BVHGPU_TUNE_FLATTEN_LAZY] != 0);   // the flat / traversal arrays carry the boxes too
}

template void refit_tree<float>(bvhgpu_tree*, const float*);
template void refit_tree<double>(bvhgpu_tree*, const double*);

}  // namespace bvhgpu
