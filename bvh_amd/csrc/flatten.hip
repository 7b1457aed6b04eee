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

template <typename T>
__global__ __launch_bounds__(256) void k_flatten(const typename Traits<T>::Node* __restrict__ nodes,
                                                 const uint32_t* __restrict__ node_start,
                                                 const uint32_t* __restrict__ node_count, const T* __restrict__ aabbs,
                                                 const uint16_t* __restrict__ node_slot, uint32_t* __restrict__ slot_entry,
                                                 typename Traits<T>::Flat* __restrict__ flat, TravNode<T>* __restrict__ trav,
                                                 uint32_t n_nodes) {
    using Tr = Traits<T>;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const typename Tr::Node nd = nodes[i];
    if (n_nodes == 1) {
        // single-shape tree: the root is a leaf and emits one leaf entry (flat_bvh.rs:129-141); its
        // traversal entry tests the shape's own AABB (flat_bvh.rs:411-418)
        typename Tr::Flat f = {};
        for (int k = 0; k < 3; k++) { f.min[k] = Tr::inf(); f.max[k] = -Tr::inf(); }
        f.entry = NONE; f.exit = 1; f.shape = nd.shape;
        flat[0] = f;
        const T* sb = aabbs + 6 * (size_t)nd.shape;
        write_trav<T>(&trav[0], sb, sb + 3, 1u, nd.shape);
        return;
    }
    if (i == 0) return;  // the root emits nothing itself (flat_bvh.rs:104-127)
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
    const uint32_t myslot = node_slot[i];
    if (myslot < TopCfg<T>::SLOTS) slot_entry[myslot] = i - 1;
    const uint32_t nav = i - 1 + L;
    typename Tr::Flat f = {};
#pragma unroll
    for (int k = 0; k < 3; k++) { f.min[k] = mn[k]; f.max[k] = mx[k]; }
    f.entry = nav + 1;
    f.exit = nav + 3 * kcnt - 1;
    f.shape = NONE;
    flat[nav] = f;
    const bool leaf = nd.shape != NONE;
    if (leaf) {
        typename Tr::Flat lf = {};
#pragma unroll
        for (int k = 0; k < 3; k++) { lf.min[k] = Tr::inf(); lf.max[k] = -Tr::inf(); }
        lf.entry = NONE; lf.exit = nav + 2; lf.shape = nd.shape;
        flat[nav + 1] = lf;
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

template <typename T> void flatten_tree(bvhgpu_tree* t) {
    using Tr = Traits<T>;
    if (t->n == 0) { t->flattened = true; return; }
    t->flat.reserve(t->n_flat * sizeof(typename Tr::Flat));
    t->trav.reserve(t->n_trav * sizeof(TravNode<T>));
    const uint32_t nn = (uint32_t)t->n_nodes;
    hipStream_t st = t->ctx->stream;
    hipLaunchKernelGGL(k_flatten<T>, dim3((nn + 255) / 256), dim3(256), 0, st,
                       t->nodes.as<typename Tr::Node>(), t->node_start.as<uint32_t>(), t->node_count.as<uint32_t>(),
                       t->aabbs.as<T>(), t->node_slot.as<uint16_t>(), t->slot_entry.as<uint32_t>(),
                       t->flat.as<typename Tr::Flat>(),
                       t->trav.as<TravNode<T>>(), nn);
    BVH_HIP(hipGetLastError());
    t->flattened = true;
}

template void flatten_tree<float>(bvhgpu_tree*);
template void flatten_tree<double>(bvhgpu_tree*);

}  // namespace bvhgpu
