// traverse.hip — <FlatBvh as BoundingHierarchy>::traverse (src/flat_bvh.rs:396-431) for a BATCH of
// rays, the triangle stage that follows it in the reference's harness (Ray::intersects_triangle,
// src/ray/ray_impl.rs:154-213; loop src/testbase.rs:826-836), Ray::new (src/ray/ray_impl.rs:70-80) and
// the bench ray stream (src/testbase.rs:687-691).
//
// A lane walks the engine's folded pre-order array (common.hpp TravNode): slab test
// (src/ray/intersect_default.rs:16-37) → hit: i+1, miss: exit.  The walk visits boxes in exactly the
// reference's order, so each ray's shapes come out in the reference's (DFS, left-first) order.
// Variable-length output (Vec<&Shape> per ray) becomes CSR in three steps:
//   1. walk: every reported shape is appended to a pool as (ray, k, shape) with k = the ray's running
//      hit count (per-wave chunks of the pool, 64 records doubling to 8192: one global atomic per chunk);
//      counts[ray] = k_end;
//   2. exclusive scan of counts → offsets (reduce + rescan; a scan of the block sums in between for > 2 M rays);
//   3. indices[offsets[ray] + k] = shape  (+ per-hit values: t-slice or triangle Intersection).
// If the pool was too small the totals are still exact; the host grows it and replays.
// Two walk kernels for the flat-array order: k_traverse (one ray per lane per launch; small or coherent batches) and
// k_traverse_lds (persistent workgroups, top of the tree resident in LDS, ray refill; large batches).  Over the BvhNode
// array: k_traverse_ordered (child-ordered iterator, LDS stack) and k_traverse_heap (best-first iterator, BinaryHeap);
// k_nearest answers nearest_to point queries.
#include <type_traits>

#include "engine.hpp"

namespace bvhgpu {

// what a walk produces besides the CSR of shape indices
enum : int {
    MODE_INDICES = 0,   // Vec<&Shape> only
    MODE_T_SLICE = 1,   // + Ray::intersection_slice_for_aabb per hit (2 scalars)
    MODE_TRIANGLES = 2, // + Ray::intersects_triangle per hit: Intersection{distance,u,v} (3 scalars)
    MODE_CLOSEST = 3    // no CSR: per ray the candidate triangle with the smallest distance
};
template <int MODE> struct ModeVals { static constexpr int N = MODE == MODE_T_SLICE ? 2 : (MODE == MODE_TRIANGLES ? 3 : 0); };

// ---- node fetch: two (f32) / four (f64) 16-byte loads per lane -------------------------------
template <typename T> struct NodeRegs { T mn[3], mx[3]; uint32_t exit, shape; };

__device__ __forceinline__ NodeRegs<float> load_node(const TravNode<float>* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    float4 a = q[0], b = q[1];
    NodeRegs<float> r;
    r.mn[0] = a.x; r.mn[1] = a.y; r.mn[2] = a.z; r.exit = __float_as_uint(a.w);
    r.mx[0] = b.x; r.mx[1] = b.y; r.mx[2] = b.z; r.shape = __float_as_uint(b.w);
    return r;
}
__device__ __forceinline__ NodeRegs<double> load_node(const TravNode<double>* p) {
    const double2* q = reinterpret_cast<const double2*>(p);
    double2 a = q[0], b = q[1], c = q[2], d = q[3];
    NodeRegs<double> r;
    r.mn[0] = a.x; r.mn[1] = a.y; r.mn[2] = b.x;
    r.mx[0] = b.y; r.mx[1] = c.x; r.mx[2] = c.y;
    unsigned long long es = (unsigned long long)__double_as_longlong(d.x);
    r.exit = (uint32_t)(es & 0xFFFFFFFFull);
    r.shape = (uint32_t)(es >> 32);
    return r;
}

struct HitRec { uint32_t ray, k, shape; };

// everything a walk kernel writes
template <typename T> struct WalkOut {
    uint32_t* counts;            // per ray: number of shapes returned
    HitRec* pool;                // hit records in arrival order
    T* pool_v;                   // ModeVals::N scalars per record
    unsigned long long pool_cap;
    unsigned long long* ctr;     // [0] pool slots taken [1] device steps [2] leaf-entry steps [4] wave steps [5] candidates (closest mode)
    const T* tris;               // n x 9 vertices (triangle modes)
    T* closest;                  // per ray {distance,u,v} (closest mode)
    uint32_t* closest_prim;      // per ray shape index or NONE
};

// ---- Ray::intersects_triangle (ray_impl.rs:154-213), Möller–Trumbore with back-face culling.  Same
//      sequence of IEEE operations as the reference: nalgebra cross = (ay*bz - az*by, az*bx - ax*bz,
//      ax*by - ay*bx), dot = (a0*b0 + a1*b1) + a2*b2, no contraction.  Returns the Intersection fields.
template <typename T> __device__ __forceinline__ void cross3(const T a[3], const T b[3], T out[3]) {
    T p0 = a[1] * b[2], q0 = a[2] * b[1];
    T p1 = a[2] * b[0], q1 = a[0] * b[2];
    T p2 = a[0] * b[1], q2 = a[1] * b[0];
    out[0] = p0 - q0; out[1] = p1 - q1; out[2] = p2 - q2;
}
template <typename T> __device__ __forceinline__ T dot3(const T a[3], const T b[3]) {
    T x = a[0] * b[0], y = a[1] * b[1], z = a[2] * b[2];
    T s = x + y;
    return s + z;
}
template <typename T>
__device__ __forceinline__ void ray_triangle(const T o[3], const T d[3], const T* __restrict__ tri, T out[3]) {
    const T a[3] = {tri[0], tri[1], tri[2]};
    T ab[3], ac[3], uvec[3], ao[3], vvec[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { ab[k] = tri[3 + k] - a[k]; ac[k] = tri[6 + k] - a[k]; }   // :170-171
    cross3<T>(d, ac, uvec);                                                                  // :176
    const T det = dot3<T>(ab, uvec);                                                         // :181
    out[0] = Traits<T>::inf(); out[1] = 0; out[2] = 0;
    if (det < Traits<T>::eps()) return;                                                      // :186-188
    const T inv_det = (T)1 / det;                                                            // :190
#pragma unroll
    for (int k = 0; k < 3; k++) ao[k] = o[k] - a[k];                                         // :193
    const T u = dot3<T>(ao, uvec) * inv_det;                                                 // :196
    out[1] = u;
    if (!(u >= (T)0 && u <= (T)1)) return;                                                   // :199-201
    cross3<T>(ao, ab, vvec);                                                                 // :204
    const T v = dot3<T>(d, vvec) * inv_det;                                                  // :207
    out[2] = v;
    if (v < (T)0 || u + v > (T)1) return;                                                    // :209-211
    const T dist = dot3<T>(ac, vvec) * inv_det;                                              // :213
    if (dist > Traits<T>::eps()) out[0] = dist;                                              // :215-219
}

// ---- per-lane ray state
template <typename T, int MODE> struct LaneRay {
    T o[3], inv[3];
    T d[MODE >= MODE_TRIANGLES ? 3 : 1];   // direction: only the triangle stage needs it
    T best[MODE == MODE_CLOSEST ? 3 : 1];  // closest Intersection so far
    uint32_t best_prim;
    uint32_t r, cnt;
    bool fin;                              // all components finite → NaN-free slab test is exact
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 3; k++) { o[k] = 0; inv[k] = 0; }
        d[0] = 0; best[0] = 0; best_prim = NONE; r = NONE; cnt = 0; fin = true;
    }
    __device__ __forceinline__ void load(const typename Traits<T>::Ray* __restrict__ rays, uint32_t ray) {
        const typename Traits<T>::Ray* rp = rays + ray;
#pragma unroll
        for (int k = 0; k < 3; k++) { o[k] = rp->o[k]; inv[k] = rp->inv[k]; }
        if (MODE >= MODE_TRIANGLES) {
#pragma unroll
            for (int k = 0; k < 3; k++) d[k] = rp->d[k];
        }
        if (MODE == MODE_CLOSEST) { best[0] = Traits<T>::inf(); best[1] = 0; best[2] = 0; }
        best_prim = NONE; r = ray; cnt = 0;
        fin = ray_is_finite<T>(o, inv);
    }
    // the ray has left the tree: its Vec / closest hit is complete
    __device__ __forceinline__ void retire(const WalkOut<T>& w) {
        if (MODE == MODE_CLOSEST) {
            w.closest[3 * (size_t)r] = best[0]; w.closest[3 * (size_t)r + 1] = best[1]; w.closest[3 * (size_t)r + 2] = best[2];
            w.closest_prim[r] = best_prim;
        } else {
            w.counts[r] = cnt;
        }
        r = NONE;
    }
};

// Hit records go to the pool in per-wave chunks: one global atomic per chunk instead of one per wave-step with
// a hit.  One address sustains only ≈88 atomics/µs on this chip, and a hit-heavy scene (58 M hits from 10 M
// primary rays on the stand-in atrium) made the walk 8x slower with fixed 64-record chunks; a wave's chunk size
// therefore doubles with every chunk it fills (64 → 8192), which bounds the slack by the records written.
constexpr uint32_t POOL_CHUNK_MIN = 64;     // >= 64: one wave-step reports at most 64 hits
constexpr uint32_t POOL_CHUNK_MAX = 8192;
// wave-uniform cursor into this wave's current chunk of the hit pool
struct PoolCursor { unsigned long long pos = 0; uint32_t left = 0; uint32_t next = POOL_CHUNK_MIN; };

// mark the unused tail of a wave's chunk so that k_hits_scatter skips it
__device__ __forceinline__ void pool_invalidate_tail(HitRec* pool, unsigned long long pool_cap, const PoolCursor& pc, int lane) {
    for (uint32_t j = (uint32_t)lane; j < pc.left; j += WAVE)
        if (pc.pos + j < pool_cap) pool[pc.pos + j].ray = NONE;
}

// A leaf box was hit (rec) in some lanes of the wave: do what the MODE asks for with the shape.
template <typename T, int MODE>
__device__ __forceinline__ void report(bool rec, uint32_t shape, T t0, T t1, LaneRay<T, MODE>& ray, const WalkOut<T>& w,
                                       PoolCursor& pc, int lane, unsigned long long lt) {
    const unsigned long long m = __ballot(rec);
    if (!m) return;
    T vals[3] = {t0, t1, 0};
    if (MODE >= MODE_TRIANGLES && rec) ray_triangle<T>(ray.o, ray.d, w.tris + 9 * (size_t)shape, vals);
    if (MODE == MODE_CLOSEST) {
        if (rec) {
            if (vals[0] < ray.best[0]) { ray.best[0] = vals[0]; ray.best[1] = vals[1]; ray.best[2] = vals[2]; ray.best_prim = shape; }
            ray.cnt++;
        }
        return;
    }
    constexpr int NV = ModeVals<MODE>::N;
    const uint32_t h = (uint32_t)__popcll(m);
    if (h > pc.left) {   // wave-uniform: start a new chunk, invalidate what is left of the old one
        pool_invalidate_tail(w.pool, w.pool_cap, pc, lane);
        unsigned int blo = 0, bhi = 0;
        if (lane == 0) {
            unsigned long long b = atomicAdd(&w.ctr[0], (unsigned long long)pc.next);
            blo = (unsigned int)b; bhi = (unsigned int)(b >> 32);
        }
        blo = __builtin_amdgcn_readfirstlane(blo); bhi = __builtin_amdgcn_readfirstlane(bhi);
        pc.pos = ((unsigned long long)bhi << 32) | blo;
        pc.left = pc.next;
        pc.next = pc.next < POOL_CHUNK_MAX ? pc.next * 2 : POOL_CHUNK_MAX;
    }
    if (rec) {
        const unsigned long long slot = pc.pos + __popcll(m & lt);
        if (slot < w.pool_cap) {
            HitRec hr; hr.ray = ray.r; hr.k = ray.cnt; hr.shape = shape;
            w.pool[slot] = hr;
#pragma unroll
            for (int k = 0; k < NV; k++) w.pool_v[NV * slot + k] = vals[k];
        }
        ray.cnt++;
    }
    pc.pos += h; pc.left -= h;
}

template <typename T, int MODE>
__device__ __forceinline__ void walk_epilogue(const WalkOut<T>& w, PoolCursor& pc, int lane, bool stats,
                                              unsigned long long steps, unsigned long long leaf_steps,
                                              unsigned long long wsteps, unsigned long long cands) {
    if (MODE != MODE_CLOSEST) pool_invalidate_tail(w.pool, w.pool_cap, pc, lane);
    if (stats) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            steps += __shfl_down(steps, d);
            leaf_steps += __shfl_down(leaf_steps, d);
            cands += __shfl_down(cands, d);
        }
        if (lane == 0) {
            atomicAdd(&w.ctr[1], steps); atomicAdd(&w.ctr[2], leaf_steps); atomicAdd(&w.ctr[4], wsteps);
            if (MODE == MODE_CLOSEST) atomicAdd(&w.ctr[5], cands);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// one ray per lane per launch
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE, bool STATS>
__global__ __launch_bounds__(256) void k_traverse(const TravNode<T>* __restrict__ nodes, uint32_t n_trav,
                                                  const typename Traits<T>::Ray* __restrict__ rays, uint32_t n_rays,
                                                  WalkOut<T> w) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    const bool active = r < n_rays;
    LaneRay<T, MODE> ray;
    ray.clear();
    if (active) ray.load(rays, r);
    uint32_t i = active ? 0u : n_trav;
    PoolCursor pc;
    unsigned long long steps = 0, leaf_steps = 0, wsteps = 0;
    // wave-uniform: every ray of this wave is finite → the NaN-free slab test (common.hpp) is exact
    const bool fast = MODE != MODE_T_SLICE && !__any(active && !ray.fin);
    while (true) {
        const bool run = i < n_trav;
        if (!__any(run)) break;
        bool rec = false;
        uint32_t shape = NONE;
        T t0 = 0, t1 = 0;
        if (STATS) wsteps++;
        if (run) {
            const NodeRegs<T> nd = load_node(nodes + i);
            const bool hit = fast ? slab_hit_finite<T>(ray.o, ray.inv, nd.mn, nd.mx)
                                  : slab_hit<T>(ray.o, ray.inv, nd.mn, nd.mx, t0, t1);
            shape = nd.shape;
            const bool leaf = trav_is_leaf(shape);
            rec = hit && leaf;
            i = hit ? i + 1 : nd.exit;   // a leaf's exit IS i+1
            if (STATS) { steps++; leaf_steps += leaf ? 1 : 0; }
        }
        report<T, MODE>(rec, shape, t0, t1, ray, w, pc, lane, lt);
    }
    const unsigned long long cands = active ? ray.cnt : 0;
    if (active) ray.retire(w);
    walk_epilogue<T, MODE>(w, pc, lane, STATS, steps, leaf_steps, wsteps, cands);
}

// ------------------------------------------------------------------------------------------------
// Ordered traversal: Bvh::nearest_child_traverse_iterator / farthest_child_traverse_iterator
// (bvh_impl.rs:184-212, bvh/child_distance_traverse.rs) collected per ray.  The iterator is a depth-first walk
// over the BvhNode array that tests both child boxes of an inner node with intersection_slice_for_aabb and
// visits the higher-priority hit child first ((left_dist > right_dist) ^ !ASCENDING → right first, :126), the
// other afterwards; a leaf yields its shape.  One ray per lane; the iterator's 32-entry stack (:36) lives in LDS
// (entry-major, so a wave's push/pop is conflict-free).  An entry holds what the iterator would do on pop:
// nothing, "go to node X" (the rest child) or "yield shape S".  A tree deeper than 32 levels makes the reference
// index out of bounds (panic); here it raises the overflow flag.
// The same set of shapes as FlatBvh::traverse comes out (slice is Some exactly when intersects_aabb is true),
// in the iterator's order; the output modes of the other walks apply.
// ------------------------------------------------------------------------------------------------
constexpr int ORD_STACK = 32;
constexpr uint32_t ORD_NOTHING = 0xFFFFFFFFu;   // RestChild::None
constexpr uint32_t ORD_YIELD = 0x80000000u;     // | shape index: a leaf was pushed (:143-147)

template <typename T, int MODE, bool ASCENDING>
__global__ __launch_bounds__(256) void k_traverse_ordered(const typename Traits<T>::Node* __restrict__ nodes, uint32_t n_nodes,
                                                          const T* __restrict__ shape_aabbs,
                                                          const typename Traits<T>::Ray* __restrict__ rays, uint32_t n_rays,
                                                          WalkOut<T> w, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t s_stack[ORD_STACK][256];
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    const bool active = r < n_rays;
    LaneRay<T, MODE> ray;
    ray.clear();
    if (active) ray.load(rays, r);
    uint32_t node_index = 0;
    int sp = 0;
    bool has_node = false;
    if (active && n_nodes) {   // iter_initially_has_node (iter.rs:164-182): a root leaf is pre-tested with the shape's AABB
        const uint32_t rs = nodes[0].shape;
        if (rs != NONE) {
            const T* sb = shape_aabbs + 6 * (size_t)rs;
            const T mn[3] = {sb[0], sb[1], sb[2]}, mx[3] = {sb[3], sb[4], sb[5]};
            T t0, t1;
            has_node = slab_hit<T>(ray.o, ray.inv, mn, mx, t0, t1);
        } else {
            has_node = true;
        }
    }
    PoolCursor pc;
    bool ovf = false;
    while (true) {
        const bool run = has_node || sp > 0;
        if (!__any(run)) break;
        bool rec = false;
        uint32_t shape = NONE;
        if (run) {
            if (has_node) {   // move_first_priority (:88-148) + stack_push (:211-213)
                const typename Traits<T>::Node* nd = nodes + node_index;
                const uint32_t ns = nd->shape;
                uint32_t entry = ORD_NOTHING;
                if (ns != NONE) {
                    has_node = false;
                    entry = ORD_YIELD | ns;
                } else {
                    T lmn[3], lmx[3], rmn[3], rmx[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) { lmn[k] = nd->l_min[k]; lmx[k] = nd->l_max[k]; rmn[k] = nd->r_min[k]; rmx[k] = nd->r_max[k]; }
                    const uint32_t li = nd->l, ri = nd->r;
                    T ld, rd, t1;
                    const bool lh = slab_hit<T>(ray.o, ray.inv, lmn, lmx, ld, t1);   // slice is Some ⇔ hit; entry = max(tmin, 0)
                    const bool rh = slab_hit<T>(ray.o, ray.inv, rmn, rmx, rd, t1);
                    if (!lh && !rh) has_node = false;
                    else if (lh && !rh) node_index = li;
                    else if (!lh && rh) node_index = ri;
                    else if ((ld > rd) != !ASCENDING) { node_index = ri; entry = li; }   // right first, left rests (:126-131)
                    else { node_index = li; entry = ri; }
                }
                if (sp >= ORD_STACK) { ovf = true; has_node = false; sp = 0; }
                else { s_stack[sp][threadIdx.x] = entry; sp++; }
            } else {          // stack_pop (:215-229)
                sp--;
                const uint32_t entry = s_stack[sp][threadIdx.x];
                if (entry == ORD_NOTHING) {
                    has_node = false;
                } else if (entry & ORD_YIELD) {
                    shape = entry & ~ORD_YIELD;
                    rec = true;
                } else {
                    node_index = entry;   // move_rest (:152-176)
                    has_node = true;
                }
            }
        }
        report<T, MODE>(rec, shape, (T)0, (T)0, ray, w, pc, lane, lt);
    }
    if (ovf) atomicOr(overflow, 1u);
    if (active) ray.retire(w);
    walk_epilogue<T, MODE>(w, pc, lane, false, 0, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Best-first traversal: Bvh::nearest_traverse_iterator / farthest_traverse_iterator (bvh_impl.rs:145-176) =
// DistanceTraverseIterator<ASCENDING> (bvh/distance_traverse.rs:40-158) collected per ray.  A max-heap of
// (dist, node) drives the walk: pop the leader; a leaf yields its shape (:151-155); an inner node tests its left,
// then its right child box with intersection_slice_for_aabb and pushes every hit child with dist = -entry
// (ascending) or exit (descending) (:99-131).  The heap is Rust's std BinaryHeap and equal distances come out
// in whatever order ITS sifts leave, so the same sifts run here: push = append + sift_up, pop = move the last
// element to the root, walk the hole down along the greater child (the right one when left <= right) to the
// bottom, then sift_up (alloc::collections::binary_heap, sift_down_to_bottom).
// One ray per lane at a time, workgroups stride over the batch.  A lane's heap: entries [0, HEAP_LDS) in LDS
// (entry-major: conflict-free), the rest in a global workspace (entry-major over all resident lanes: coalesced
// when lanes touch the same entry).  The frontier of a best-first walk is small (peak 10 on the 120k-triangle
// scene, 15 on the atrium stand-in), so the global part is touched only by unusual rays; if even that
// overflows the host doubles it and replays.
// ------------------------------------------------------------------------------------------------
constexpr int HEAP_LDS = 16;
constexpr uint32_t HEAP_OVERFLOW_BIT = 2u;

template <typename T> struct LaneHeap {
    T (*sd)[256];
    uint32_t (*sn)[256];
    T* gd;
    uint32_t* gn;
    size_t G, g;
    uint32_t tid;
    __device__ __forceinline__ T dist(uint32_t e) const { return e < HEAP_LDS ? sd[e][tid] : gd[(size_t)(e - HEAP_LDS) * G + g]; }
    __device__ __forceinline__ uint32_t node(uint32_t e) const { return e < HEAP_LDS ? sn[e][tid] : gn[(size_t)(e - HEAP_LDS) * G + g]; }
    __device__ __forceinline__ void put(uint32_t e, T d, uint32_t n) {
        if (e < HEAP_LDS) { sd[e][tid] = d; sn[e][tid] = n; }
        else { gd[(size_t)(e - HEAP_LDS) * G + g] = d; gn[(size_t)(e - HEAP_LDS) * G + g] = n; }
    }
    // BinaryHeap::sift_up(0, pos) with the element held in registers (the std's Hole)
    __device__ __forceinline__ void sift_up(uint32_t pos, T d, uint32_t n) {
        while (pos > 0) {
            const uint32_t parent = (pos - 1) >> 1;
            const T pd = dist(parent);
            if (d <= pd) break;
            put(pos, pd, node(parent));
            pos = parent;
        }
        put(pos, d, n);
    }
};

template <typename T, int MODE, bool ASCENDING>
__global__ __launch_bounds__(256) void k_traverse_heap(const typename Traits<T>::Node* __restrict__ nodes, uint32_t n_nodes,
                                                       const T* __restrict__ shape_aabbs,
                                                       const typename Traits<T>::Ray* __restrict__ rays, uint32_t n_rays,
                                                       WalkOut<T> w, T* __restrict__ heap_dist, uint32_t* __restrict__ heap_node,
                                                       uint32_t heap_cap, uint32_t* __restrict__ overflow) {
    __shared__ T s_dist[HEAP_LDS][256];
    __shared__ uint32_t s_node[HEAP_LDS][256];
    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    LaneHeap<T> hp;
    hp.sd = s_dist; hp.sn = s_node; hp.gd = heap_dist; hp.gn = heap_node;
    hp.G = (size_t)gridDim.x * 256; hp.g = (size_t)blockIdx.x * 256 + threadIdx.x; hp.tid = threadIdx.x;
    const uint32_t cap = HEAP_LDS + heap_cap;
    PoolCursor pc;
    bool ovf = false;
    LaneRay<T, MODE> ray;
    for (size_t base = (size_t)blockIdx.x * 256; base < n_rays; base += hp.G) {   // workgroup-uniform
        const size_t r = base + threadIdx.x;
        const bool active = r < n_rays;
        ray.clear();
        if (active) ray.load(rays, (uint32_t)r);
        uint32_t len = 0;
        if (active && n_nodes) {   // iter_initially_has_node (iter.rs:164-182), then add_to_heap(T::zero(), 0) (:75-78)
            bool has_node = true;
            const uint32_t rs = nodes[0].shape;
            if (rs != NONE) {
                const T* sb = shape_aabbs + 6 * (size_t)rs;
                const T mn[3] = {sb[0], sb[1], sb[2]}, mx[3] = {sb[3], sb[4], sb[5]};
                T t0, t1;
                has_node = slab_hit<T>(ray.o, ray.inv, mn, mx, t0, t1);
            }
            if (has_node) { hp.put(0, ASCENDING ? -(T)0 : (T)0, 0u); len = 1; }
        }
        while (true) {
            const bool run = len > 0;
            if (!__any(run)) break;
            bool rec = false;
            uint32_t shape = NONE;
            if (run) {
                // BinaryHeap::pop
                len--;
                const T last_d = hp.dist(len);
                uint32_t node_index = hp.node(len);
                if (len > 0) {
                    const uint32_t last_n = node_index;
                    node_index = hp.node(0);
                    uint32_t pos = 0, child = 1;
                    while (child + 1 < len) {            // child <= end.saturating_sub(2)
                        T cd = hp.dist(child);
                        const T cr = hp.dist(child + 1);
                        if (cd <= cr) { child++; cd = cr; }
                        hp.put(pos, cd, hp.node(child));
                        pos = child;
                        child = 2 * pos + 1;
                    }
                    if (child == len - 1) { hp.put(pos, hp.dist(child), hp.node(child)); pos = child; }
                    hp.sift_up(pos, last_d, last_n);
                }
                // unpack_node (:82-97)
                const typename Traits<T>::Node* nd = nodes + node_index;
                const uint32_t ns = nd->shape;
                if (ns != NONE) {
                    rec = true; shape = ns;
                } else {
                    T lmn[3], lmx[3], rmn[3], rmx[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) { lmn[k] = nd->l_min[k]; lmx[k] = nd->l_max[k]; rmn[k] = nd->r_min[k]; rmx[k] = nd->r_max[k]; }
                    const uint32_t li = nd->l, ri = nd->r;
                    T l0, l1, r0, r1;
                    const bool lh = slab_hit<T>(ray.o, ray.inv, lmn, lmx, l0, l1);   // slice is Some ⇔ hit: (max(tmin,0), tmax)
                    const bool rh = slab_hit<T>(ray.o, ray.inv, rmn, rmx, r0, r1);
                    if (len + (lh ? 1u : 0u) + (rh ? 1u : 0u) > cap) {
                        ovf = true; len = 0;             // the host grows the workspace and replays the batch
                    } else {
                        if (lh) { hp.sift_up(len, ASCENDING ? -l0 : l1, li); len++; }   // BinaryHeap::push
                        if (rh) { hp.sift_up(len, ASCENDING ? -r0 : r1, ri); len++; }
                    }
                }
            }
            report<T, MODE>(rec, shape, (T)0, (T)0, ray, w, pc, lane, lt);
        }
        if (active) ray.retire(w);
    }
    if (ovf) atomicOr(overflow, HEAP_OVERFLOW_BIT);
    walk_epilogue<T, MODE>(w, pc, lane, false, 0, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// LDS-resident top of the tree.  On the 120k-triangle scene 72 % of all box tests touch the first 11
// levels of the tree (2047 entries) and the vector L1 — one tag lookup per lane per 16-byte load for
// these scattered reads — is the unit that saturates (measured: ~1 lane-access per clock per CU).  A
// 1024-thread workgroup copies the entries whose heap number is below K into LDS (split into 16-byte
// planes so that a ds_read_b128 of 16 lanes spreads over all 16 bank quads; 2 workgroups of 64 KB per
// CU) and every lane tracks the slot of its current entry: descend → 2*slot, miss → the exit's slot
// carried in the entry's spare word.  A lane outside the resident set (deep in the tree, or after a
// leaf) reads L2 as before and re-enters the resident set through the same word.
// The workgroup's waves draw rays from ONE cursor in LDS (a wave-aggregated ds_add per refill phase), so
// the tail of a launch is the tail of a workgroup's ray range, not of every wave's.  Retiring and
// refilling lanes is kept OUT of the walk loop: LDS_INNER lean steps (~38 VALU each), then one refill
// phase; a lane whose ray ends mid-way idles for at most LDS_INNER-1 steps.
// ------------------------------------------------------------------------------------------------
template <typename T> struct TopLds;
template <> struct TopLds<float> {
    static constexpr uint32_t BYTES_PER_SLOT = 32;
    float4 *lo, *hi;
    __device__ __forceinline__ TopLds(unsigned char* base, uint32_t K) {
        lo = reinterpret_cast<float4*>(base); hi = lo + K;
    }
    __device__ __forceinline__ void store(uint32_t q, const TravNode<float>* g) {
        const float4* p = reinterpret_cast<const float4*>(g);
        lo[q] = p[0]; hi[q] = p[1];
    }
    __device__ __forceinline__ NodeRegs<float> load(uint32_t q) const {
        const float4 a = lo[q], b = hi[q];
        NodeRegs<float> r;
        r.mn[0] = a.x; r.mn[1] = a.y; r.mn[2] = a.z; r.exit = __float_as_uint(a.w);
        r.mx[0] = b.x; r.mx[1] = b.y; r.mx[2] = b.z; r.shape = __float_as_uint(b.w);
        return r;
    }
};
template <> struct TopLds<double> {
    static constexpr uint32_t BYTES_PER_SLOT = 56;
    double2 *a, *b, *c;
    uint2* d;
    __device__ __forceinline__ TopLds(unsigned char* base, uint32_t K) {
        a = reinterpret_cast<double2*>(base); b = a + K; c = b + K; d = reinterpret_cast<uint2*>(c + K);
    }
    __device__ __forceinline__ void store(uint32_t q, const TravNode<double>* g) {
        const double2* p = reinterpret_cast<const double2*>(g);
        a[q] = p[0]; b[q] = p[1]; c[q] = p[2];
        const unsigned long long es = (unsigned long long)__double_as_longlong(p[3].x);
        d[q] = make_uint2((uint32_t)(es & 0xFFFFFFFFull), (uint32_t)(es >> 32));
    }
    __device__ __forceinline__ NodeRegs<double> load(uint32_t q) const {
        const double2 x = a[q], y = b[q], z = c[q];
        const uint2 w = d[q];
        NodeRegs<double> r;
        r.mn[0] = x.x; r.mn[1] = x.y; r.mn[2] = y.x;
        r.mx[0] = y.y; r.mx[1] = z.x; r.mx[2] = z.y;
        r.exit = w.x; r.shape = w.y;
        return r;
    }
};

constexpr int LDS_THREADS = 1024;
#ifndef BVH_LDS_INNER
#define BVH_LDS_INNER 8
#endif
constexpr int LDS_INNER = BVH_LDS_INNER;   // walk steps between two refill phases (4 / 6 / 8 / 10 / 12 / 16 measured: 8)

template <typename T, int MODE, bool STATS>
__global__ __launch_bounds__(LDS_THREADS) void k_traverse_lds(const TravNode<T>* __restrict__ nodes, uint32_t n_trav,
                                                               const uint32_t* __restrict__ slot_entry, uint32_t K,
                                                               uint32_t first_slot, uint32_t split,
                                                               const typename Traits<T>::Ray* __restrict__ rays,
                                                               uint32_t n_rays, uint32_t rays_per_wg, WalkOut<T> w) {
    // split != 0: every ray is walked as TWO independent items, item 2r over the entries of the root's left
    // subtree [0, split_at) and item 2r+1 over the right one [split_at, n_trav).  The per-ray list is the
    // concatenation of the two (pre-order!), so the CSR machinery simply runs over 2R items.  At 1 M rays a lane
    // only gets ~2 rays; halving the longest walks and doubling the items per lane shortens the tail of the launch.
    // (n_rays and rays_per_wg count items here.)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t& s_next = *reinterpret_cast<uint32_t*>(smem);
    TopLds<T> top(smem + 16, K);
    const uint32_t split_at = split ? load_node(nodes).exit : 0u;   // wave-uniform
    const unsigned long long g0 = (unsigned long long)blockIdx.x * rays_per_wg;
    const unsigned long long g1 = g0 + rays_per_wg;
    const uint32_t wg_begin = (uint32_t)(g0 < n_rays ? g0 : n_rays);
    const uint32_t wg_end = (uint32_t)(g1 < n_rays ? g1 : n_rays);
    if (threadIdx.x == 0) s_next = wg_begin;
    for (uint32_t q = threadIdx.x; q < K; q += blockDim.x) {
        const uint32_t e = slot_entry[q];
        if (e != NONE) top.store(q, nodes + e);
    }
    __syncthreads();

    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    LaneRay<T, MODE> ray;
    ray.clear();
    uint32_t i = 0, limit = 0, slot = SLOT_NONE;   // the walk runs while i < limit
    bool exhausted = wg_begin >= wg_end;   // wave-uniform: the workgroup's range has been handed out
    PoolCursor pc;
    unsigned long long steps = 0, leaf_steps = 0, wsteps = 0, cands = 0;
    while (true) {
        // ---- refill phase
        bool run = i < limit;
        const unsigned long long idle = __ballot(!run);
        if (idle) {
            if (!run && ray.r != NONE) { cands += ray.cnt; ray.retire(w); }
            if (!exhausted) {
                const uint32_t nidle = (uint32_t)__popcll(idle);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_next, nidle);
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t mine = base + (uint32_t)__popcll(idle & lt);
                if (!run && base < wg_end && mine < wg_end) {
                    if (split_at) {
                        const bool right = (mine & 1u) != 0u;
                        ray.load(rays, mine >> 1);
                        ray.r = mine;                       // counts / pool records are per item
                        i = right ? split_at : 0u; limit = right ? n_trav : split_at;
                        slot = right ? 3u : 2u;             // heap numbers of the root's children
                    } else {
                        ray.load(rays, mine);
                        i = 0; limit = n_trav; slot = first_slot;
                    }
                    run = true;
                }
                exhausted = base >= wg_end || (wg_end - base) <= nidle;
            }
            if (!__any(run)) break;
        }
        const bool fast = MODE != MODE_T_SLICE && !__any(run && !ray.fin);   // wave-uniform
        // ---- LDS_INNER walk steps
        for (int s = 0; s < LDS_INNER; s++) {
            bool rec = false;
            uint32_t shape = NONE;
            T t0 = 0, t1 = 0;
            if (STATS) wsteps++;
            if (i < limit) {
                NodeRegs<T> nd;
                if (slot < K) nd = top.load(slot);
                else nd = load_node(nodes + i);
                const bool hit = fast ? slab_hit_finite<T>(ray.o, ray.inv, nd.mn, nd.mx)
                                      : slab_hit<T>(ray.o, ray.inv, nd.mn, nd.mx, t0, t1);
                shape = nd.shape;
                const bool leaf = trav_is_leaf(shape);
                rec = hit && leaf;
                const bool descend = hit && !leaf;
                i = descend ? i + 1 : nd.exit;   // a leaf's exit IS i+1
                const uint32_t child = min(slot << 1, SLOT_NONE);
                slot = descend ? child : (leaf ? SLOT_NONE : (shape & 0xFFFFu));
                if (STATS) { steps++; leaf_steps += leaf ? 1 : 0; }
            }
            report<T, MODE>(rec, shape, t0, t1, ray, w, pc, lane, lt);
        }
    }
    walk_epilogue<T, MODE>(w, pc, lane, STATS, steps, leaf_steps, wsteps, cands);
}

// ---- exclusive scan of per-ray counts ----------------------------------------------------------
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;

// PAIR: every ray was walked as two items (k_traverse_lds split): its count is counts[2r] + counts[2r+1]
template <bool PAIR> __device__ __forceinline__ uint32_t ray_count(const uint32_t* __restrict__ counts, uint32_t r) {
    if (!PAIR) return counts[r];
    const uint2 c = reinterpret_cast<const uint2*>(counts)[r];
    return c.x + c.y;
}

template <bool PAIR>
__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ counts, uint32_t n,
                                                     unsigned long long* __restrict__ blocksums) {
    __shared__ unsigned long long ws[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += (base + j < n) ? ray_count<PAIR>(counts, base + j) : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if (lane_id() == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) blocksums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void k_scan_sums(unsigned long long* __restrict__ blocksums, uint32_t nb,
                                                   unsigned long long* __restrict__ total_out) {
    __shared__ unsigned long long sh[256];
    unsigned long long carry = 0;
    for (uint32_t c0 = 0; c0 < nb; c0 += 256) {
        const uint32_t j = c0 + threadIdx.x;
        const unsigned long long v = j < nb ? blocksums[j] : 0ull;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            unsigned long long u = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0ull;
            __syncthreads();
            sh[threadIdx.x] += u;
            __syncthreads();
        }
        if (j < nb) blocksums[j] = carry + sh[threadIdx.x] - v;
        carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

constexpr uint32_t SCAN_FUSED_MAX_BLOCKS = 2048;   // up to this many blocks every block sums its predecessors itself

// counts → offsets.  PREFIXED: blocksums already hold exclusive prefixes (k_scan_sums ran, large batches); otherwise
// they are the raw per-block sums of k_scan_reduce and this block adds up its predecessors (one kernel less).
template <bool PAIR, bool PREFIXED>
__global__ __launch_bounds__(256) void k_scan_final(const uint32_t* __restrict__ counts, uint32_t n,
                                                    const unsigned long long* __restrict__ blocksums,
                                                    unsigned long long* __restrict__ total,
                                                    uint32_t* __restrict__ offsets) {
    __shared__ uint32_t ws[4];
    __shared__ unsigned long long wb[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    const int lane = lane_id();
    unsigned long long before = 0;
    if (PREFIXED) {
        before = blocksums[blockIdx.x];
    } else {
        unsigned long long part = 0;
        for (uint32_t j = threadIdx.x; j < blockIdx.x; j += 256) part += blocksums[j];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d);
        if (lane == 0) wb[threadIdx.x >> 6] = part;
    }
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) { v[j] = (base + j < n) ? ray_count<PAIR>(counts, base + j) : 0u; s += v[j]; }
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t u = __shfl_up(inc, d);
        if (lane >= d) inc += u;
    }
    if (lane == WAVE - 1) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (!PREFIXED) before = wb[0] + wb[1] + wb[2] + wb[3];
    uint32_t wbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wbase += ws[w];
    uint32_t run = (uint32_t)before + wbase + inc - s;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        if (base + j < n) offsets[base + j] = run;
        run += v[j];
    }
    if (PREFIXED) {
        if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n] = (uint32_t)(*total);
    } else if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {   // the last block knows the total
        const unsigned long long t = before + ws[0] + ws[1] + ws[2] + ws[3];
        offsets[n] = (uint32_t)t;
        *total = t;
    }
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void k_hits_scatter(const HitRec* __restrict__ pool, const T* __restrict__ pool_v,
                                                      const unsigned long long* __restrict__ ctr,
                                                      unsigned long long pool_cap, const uint32_t* __restrict__ offsets,
                                                      const uint32_t* __restrict__ pair_counts,
                                                      uint32_t* __restrict__ indices, T* __restrict__ vals) {
    const unsigned long long n = ctr[0];
    if (n > pool_cap) return;  // pool overflowed: indices[] is too small as well; the host grows both and replays
    for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < n;
         j += (unsigned long long)gridDim.x * blockDim.x) {
        const HitRec h = pool[j];
        if (h.ray == NONE) continue;   // unused tail of a per-wave chunk
        // pair_counts: h.ray is an ITEM (2*ray + side); the right item's records follow the left item's
        const uint32_t d = pair_counts ? offsets[h.ray >> 1] + ((h.ray & 1u) ? pair_counts[h.ray - 1] : 0u) + h.k
                                       : offsets[h.ray] + h.k;
        indices[d] = h.shape;
#pragma unroll
        for (int k = 0; k < NV; k++) vals[NV * (size_t)d + k] = pool_v[NV * j + k];
    }
}

// The 8 walk / scan counters go to the context's pinned host page and are zeroed for the next call: one 64-thread
// launch instead of the runtime's copy kernel plus its fill kernel (≈4.5 µs each on the stream).
__global__ void k_publish_counters(unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ host_page) {
    if (threadIdx.x < 8) {
        host_page[threadIdx.x] = ctr[threadIdx.x];
        ctr[threadIdx.x] = 0;
    }
    __threadfence_system();
}

// ------------------------------------------------------------------------------------------------
template <typename T, int MODE, bool STATS>
static void launch_walk(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, const WalkOut<T>& w, bool use_lds,
                        uint32_t split_at) {
    bvhgpu_ctx* ctx = t->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t n_trav = (uint32_t)t->n_trav;
    const TravNode<T>* nodes = t->trav.as<TravNode<T>>();
    if (!use_lds) {   // one ray per lane per launch
        hipLaunchKernelGGL((k_traverse<T, MODE, STATS>), dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, nodes,
                           n_trav, rays_dev, (uint32_t)n_rays, w);
        return;
    }
    // workgroups of lds_threads that each keep K top-of-tree slots in LDS; as many per CU as 160 KB of LDS
    // and 32 waves allow
    // 0 = per-type default: as many slots as let TWO workgroups share a CU's 160 KB (f32: 2559 x 32 B, f64: 1462 x 56 B) — one
    // slot more halves the occupancy (0.206 → 0.296 ms on configs[1]); 1024 threads for f32, 512 for f64 (tools/f64_sweep.py)
    const bool wide = sizeof(T) == 8;
    const int two_per_cu = (int)(((160 * 1024) / 2 - 16) / TopLds<T>::BYTES_PER_SLOT);
    const int want_threads = ctx->tune[BVHGPU_TUNE_TRAVERSE_LDS_THREADS] > 0 ? ctx->tune[BVHGPU_TUNE_TRAVERSE_LDS_THREADS] : (wide ? 512 : 1024);
    const int want_slots = ctx->tune[BVHGPU_TUNE_TRAVERSE_LDS_SLOTS] > 0 ? ctx->tune[BVHGPU_TUNE_TRAVERSE_LDS_SLOTS] : two_per_cu;
    const uint32_t lds_threads = (uint32_t)std::min(LDS_THREADS, std::max(64, want_threads & ~63));
    const uint32_t K = (uint32_t)std::min<int>((int)TopCfg<T>::SLOTS, std::max(4, want_slots));
    const size_t lds_bytes = 16 + (size_t)K * TopLds<T>::BYTES_PER_SLOT;
    const uint32_t wg_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds_bytes, 2048 / lds_threads));
    const size_t n_items = split_at ? 2 * n_rays : n_rays;
    const size_t full = (n_items + WAVE - 1) / WAVE;
    const uint32_t n_waves = (uint32_t)std::min<size_t>(full, (size_t)ctx->n_cu * wg_per_cu * (lds_threads / WAVE));
    const dim3 lgrid((n_waves + lds_threads / WAVE - 1) / (lds_threads / WAVE));
    const uint32_t rpg = (uint32_t)((n_items + lgrid.x - 1) / lgrid.x);   // items per workgroup
    const uint32_t first_slot = t->n >= 2 ? 2u : SLOT_NONE;               // entry 0 is the root's left child (heap number 2)
    static thread_local size_t lds_attr[16] = {};   // per device: dynamic-LDS limit already set for this instantiation
    size_t& have = lds_attr[ctx->device & 15];
    if (have < lds_bytes) {
        BVH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_traverse_lds<T, MODE, STATS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        have = lds_bytes;
    }
    hipLaunchKernelGGL((k_traverse_lds<T, MODE, STATS>), lgrid, dim3(lds_threads), lds_bytes, st, nodes, n_trav,
                       t->slot_entry.as<uint32_t>(), K, first_slot, split_at, rays_dev, (uint32_t)n_items, rpg, w);
}

template <typename T>
void traverse_batch(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, unsigned flags,
                    bvhgpu_hits* h) {
    bvhgpu_ctx* ctx = t->ctx;
    hipStream_t st = ctx->stream;
    const bool stats = (flags & BVHGPU_TRAVERSE_STATS) != 0;
    const bool coherent = (flags & BVHGPU_TRAVERSE_COHERENT) != 0;
    const int ordered = (flags & BVHGPU_TRAVERSE_NEAREST_FIRST) ? 1 : ((flags & BVHGPU_TRAVERSE_FARTHEST_FIRST) ? 2 : 0);
    const int mode = (flags & BVHGPU_TRAVERSE_CLOSEST) ? MODE_CLOSEST
                   : (flags & BVHGPU_TRAVERSE_TRIANGLES) ? MODE_TRIANGLES
                   : (flags & BVHGPU_TRAVERSE_T_SLICE) ? MODE_T_SLICE : MODE_INDICES;
    const int nv = mode == MODE_T_SLICE ? 2 : (mode == MODE_TRIANGLES ? 3 : 0);
    // walk kernel: one ray per lane per launch, or persistent workgroups with the top of the tree in LDS
    const bool use_lds = ctx->tune[BVHGPU_TUNE_TRAVERSE_VARIANT] != 0 && t->slot_entry.p != nullptr && !coherent && !ordered &&
                         n_rays >= (size_t)ctx->tune[BVHGPU_TUNE_TRAVERSE_LDS_MIN_RAYS];
    // two items per ray (left / right subtree of the root) for the CSR modes: see k_traverse_lds
    uint32_t split_at = 0;
    if (use_lds && mode != MODE_CLOSEST && ctx->tune[BVHGPU_TUNE_TRAVERSE_SPLIT] != 0 && t->n >= 2 && !t->unfolded &&
        n_rays < (size_t)ctx->n_cu * 2048 * 4) {   // fewer than 4 rays per resident lane: the tail dominates
        split_at = 1;   // the kernel reads the boundary itself: exit index of entry 0 (the root's left child)
    }
    const size_t n_items = split_at ? 2 * n_rays : n_rays;
    h->ctx = ctx; h->dtype = Traits<T>::dtype; h->n_rays = n_rays; h->flags = flags; h->total = 0;
    h->stats = bvhgpu_traverse_stats{0, 0, 0, 0, 0};
    { const void* before = h->ctr.p; h->ctr.reserve(8 * sizeof(unsigned long long)); if (h->ctr.p != before) h->ctr_clean = false; }
    unsigned long long* pin = reinterpret_cast<unsigned long long*>(ctx->pinned);
    unsigned long long* ctr = h->ctr.as<unsigned long long>();

    WalkOut<T> w;
    w.counts = nullptr; w.pool = nullptr; w.pool_v = nullptr; w.pool_cap = 0; w.ctr = ctr;
    w.tris = t->tris.as<T>(); w.closest = nullptr; w.closest_prim = nullptr;

    uint32_t* ovf_flag = reinterpret_cast<uint32_t*>(ctr + 7);   // ordered walk: iterator stack (bit 0) / heap workspace (bit 1) overflow
    const bool best_first = ordered && (flags & BVHGPU_TRAVERSE_BEST_FIRST) != 0;
    const unsigned heap_grid = (unsigned)std::min<size_t>((n_rays + 255) / 256, (size_t)ctx->n_cu * 4);
    auto launch_ordered = [&](auto mode_tag, auto asc_tag) {
        constexpr int M = decltype(mode_tag)::value;
        constexpr bool A = decltype(asc_tag)::value;
        if (best_first) {   // DistanceTraverseIterator
            const size_t lanes = (size_t)heap_grid * 256;
            if (lanes * h->heap_cap * (sizeof(T) + 4) > ((size_t)16 << 30))
                throw HipFail{hipErrorInvalidValue, "ORDERED_DEPTH", __LINE__};
            h->heap_dist.reserve(lanes * h->heap_cap * sizeof(T));
            h->heap_node.reserve(lanes * h->heap_cap * 4);
            hipLaunchKernelGGL((k_traverse_heap<T, M, A>), dim3(heap_grid), dim3(256), 0, st,
                               t->nodes.as<typename Traits<T>::Node>(), (uint32_t)t->n_nodes, t->aabbs.as<T>(), rays_dev,
                               (uint32_t)n_rays, w, h->heap_dist.as<T>(), h->heap_node.as<uint32_t>(), h->heap_cap, ovf_flag);
            return;
        }
        hipLaunchKernelGGL((k_traverse_ordered<T, M, A>), dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st,
                           t->nodes.as<typename Traits<T>::Node>(), (uint32_t)t->n_nodes, t->aabbs.as<T>(), rays_dev,
                           (uint32_t)n_rays, w, ovf_flag);
    };
    auto dispatch_ordered = [&](auto mode_tag) {
        if (ordered == 1) launch_ordered(mode_tag, std::true_type{}); else launch_ordered(mode_tag, std::false_type{});
    };
#define DISPATCH_WALK()                                                                              \
    do {                                                                                             \
        if (ordered) {                                                                               \
            switch (mode) {                                                                          \
                case MODE_INDICES: dispatch_ordered(std::integral_constant<int, MODE_INDICES>{}); break;     \
                case MODE_TRIANGLES: dispatch_ordered(std::integral_constant<int, MODE_TRIANGLES>{}); break; \
                default: dispatch_ordered(std::integral_constant<int, MODE_CLOSEST>{}); break;       \
            }                                                                                        \
            break;                                                                                   \
        }                                                                                            \
        switch (mode) {                                                                              \
            case MODE_INDICES: if (stats) launch_walk<T, MODE_INDICES, true>(t, rays_dev, n_rays, w, use_lds, split_at);  \
                               else launch_walk<T, MODE_INDICES, false>(t, rays_dev, n_rays, w, use_lds, split_at); break; \
            case MODE_T_SLICE: if (stats) launch_walk<T, MODE_T_SLICE, true>(t, rays_dev, n_rays, w, use_lds, split_at);  \
                               else launch_walk<T, MODE_T_SLICE, false>(t, rays_dev, n_rays, w, use_lds, split_at); break; \
            case MODE_TRIANGLES: if (stats) launch_walk<T, MODE_TRIANGLES, true>(t, rays_dev, n_rays, w, use_lds, split_at); \
                                 else launch_walk<T, MODE_TRIANGLES, false>(t, rays_dev, n_rays, w, use_lds, split_at); break; \
            default: if (stats) launch_walk<T, MODE_CLOSEST, true>(t, rays_dev, n_rays, w, use_lds, split_at);           \
                     else launch_walk<T, MODE_CLOSEST, false>(t, rays_dev, n_rays, w, use_lds, split_at); break;         \
        }                                                                                            \
    } while (0)

    if (mode == MODE_CLOSEST) {   // no CSR: one Intersection + shape per ray
        h->closest.reserve(std::max<size_t>(n_rays, 1) * 3 * sizeof(T));
        h->closest_prim.reserve(std::max<size_t>(n_rays, 1) * 4);
        if (n_rays == 0) return;
        w.closest = h->closest.as<T>(); w.closest_prim = h->closest_prim.as<uint32_t>();
        for (;;) {
            if (!h->ctr_clean) BVH_HIP(hipMemsetAsync(ctr, 0, 8 * sizeof(unsigned long long), st));
            h->ctr_clean = false;
            if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[4], st)); }
            DISPATCH_WALK();
            if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[5], st)); BVH_HIP(hipEventRecord(ctx->ev[6], st)); }
            hipLaunchKernelGGL(k_publish_counters, dim3(1), dim3(64), 0, st, ctr, pin);   // readback + reset for the next call
            BVH_HIP(hipStreamSynchronize(st));
            h->ctr_clean = true;
            BVH_HIP(hipGetLastError());
            if (best_first && (pin[7] & HEAP_OVERFLOW_BIT)) { h->heap_cap *= 2; continue; }   // a lane's heap outgrew the workspace
            break;
        }
        if (ordered && (pin[7] & 1ull)) throw HipFail{hipErrorInvalidValue, "ORDERED_DEPTH", __LINE__};
        if (stats) {
            const bool one_to_one = t->unfolded || t->n == 1;
            h->stats.hits = pin[5];
            h->stats.device_steps = pin[1];
            h->stats.wave_steps = pin[4];
            h->stats.visited = one_to_one ? pin[1] : pin[1] + pin[5];
            h->stats.leaf_visits = one_to_one ? pin[2] : pin[5];
        }
        if (ctx->timing) ctx->ev_set |= 4u;
        return;
    }

    h->counts.reserve((n_items + 1) * 4);
    h->offsets.reserve((n_rays + 1) * 4);
    const uint32_t nb = (uint32_t)((n_rays + SCAN_BLOCK - 1) / SCAN_BLOCK);
    h->blocksums.reserve((nb + 1) * sizeof(unsigned long long));
    if (h->pool_cap == 0) h->pool_cap = std::max<size_t>(n_rays, (size_t)1 << 16);
    if (n_rays == 0) {
        BVH_HIP(hipMemsetAsync(h->offsets.p, 0, 4, st));
        BVH_HIP(hipStreamSynchronize(st));
        return;
    }
    for (int attempt = 0; attempt < (best_first ? 24 : 2); attempt++) {
        h->pool.reserve(h->pool_cap * sizeof(HitRec));
        h->indices.reserve(h->pool_cap * 4);
        if (nv) {
            h->pool_t.reserve(h->pool_cap * nv * sizeof(T));
            (mode == MODE_T_SLICE ? h->tslice : h->isect).reserve(h->pool_cap * nv * sizeof(T));
        }
        if (!h->ctr_clean) BVH_HIP(hipMemsetAsync(ctr, 0, 8 * sizeof(unsigned long long), st));
        h->ctr_clean = false;
        const unsigned long long cap = h->pool_cap;
        w.counts = h->counts.as<uint32_t>(); w.pool = h->pool.as<HitRec>(); w.pool_v = h->pool_t.as<T>(); w.pool_cap = cap;
        uint32_t* counts = w.counts;
        if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[4], st)); }
        DISPATCH_WALK();
        if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[5], st)); }
        unsigned long long* bs = h->blocksums.as<unsigned long long>();
        uint32_t* offs = h->offsets.as<uint32_t>();
        if (split_at) hipLaunchKernelGGL(k_scan_reduce<true>, dim3(nb), dim3(256), 0, st, counts, (uint32_t)n_rays, bs);
        else hipLaunchKernelGGL(k_scan_reduce<false>, dim3(nb), dim3(256), 0, st, counts, (uint32_t)n_rays, bs);
        if (nb > SCAN_FUSED_MAX_BLOCKS) {
            hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, bs, nb, ctr + 3);
            if (split_at) hipLaunchKernelGGL((k_scan_final<true, true>), dim3(nb), dim3(256), 0, st, counts, (uint32_t)n_rays, bs, ctr + 3, offs);
            else hipLaunchKernelGGL((k_scan_final<false, true>), dim3(nb), dim3(256), 0, st, counts, (uint32_t)n_rays, bs, ctr + 3, offs);
        } else {
            if (split_at) hipLaunchKernelGGL((k_scan_final<true, false>), dim3(nb), dim3(256), 0, st, counts, (uint32_t)n_rays, bs, ctr + 3, offs);
            else hipLaunchKernelGGL((k_scan_final<false, false>), dim3(nb), dim3(256), 0, st, counts, (uint32_t)n_rays, bs, ctr + 3, offs);
        }
        const uint32_t* pair_counts = split_at ? counts : nullptr;
        const int sgrid = (int)std::min<size_t>((cap + 255) / 256, (size_t)ctx->n_cu * 8);
        T* vals = mode == MODE_T_SLICE ? h->tslice.as<T>() : h->isect.as<T>();
        if (nv == 2)
            hipLaunchKernelGGL((k_hits_scatter<T, 2>), dim3(sgrid), dim3(256), 0, st, w.pool, w.pool_v, ctr, cap,
                               offs, pair_counts, h->indices.as<uint32_t>(), vals);
        else if (nv == 3)
            hipLaunchKernelGGL((k_hits_scatter<T, 3>), dim3(sgrid), dim3(256), 0, st, w.pool, w.pool_v, ctr, cap,
                               offs, pair_counts, h->indices.as<uint32_t>(), vals);
        else
            hipLaunchKernelGGL((k_hits_scatter<T, 0>), dim3(sgrid), dim3(256), 0, st, w.pool, w.pool_v, ctr, cap,
                               offs, pair_counts, h->indices.as<uint32_t>(), vals);
        if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[6], st)); }
        hipLaunchKernelGGL(k_publish_counters, dim3(1), dim3(64), 0, st, ctr, pin);   // readback + reset for the next call
        BVH_HIP(hipStreamSynchronize(st));
        h->ctr_clean = true;
        BVH_HIP(hipGetLastError());
        if (best_first && (pin[7] & HEAP_OVERFLOW_BIT)) { h->heap_cap *= 2; continue; }   // a lane's heap outgrew the workspace
        if (ordered && (pin[7] & 1ull)) throw HipFail{hipErrorInvalidValue, "ORDERED_DEPTH", __LINE__};
        const unsigned long long used = pin[0];   // pool slots taken (whole chunks)
        const unsigned long long total = pin[3];  // sum of the per-ray counts = number of hits
        if (used < total) throw HipFail{hipErrorUnknown, "hit pool / count scan mismatch", __LINE__};
        if (total > 0xFFFFFFFFull) throw HipFail{hipErrorInvalidValue, "OVERFLOW", __LINE__};
        if (used > cap) {  // pool too small: grow to the need (deterministic: same chunks on replay) and replay
            h->pool_cap = (size_t)used + (size_t)used / 8 + 1024;
            continue;
        }
        h->total = total;
        h->stats.hits = total;
        if (stats) {
            h->stats.device_steps = pin[1];
            h->stats.wave_steps = pin[4];
            // reference-equivalent loop iterations (flat_bvh.rs:408): in the folded layout every
            // reported leaf stands for a navigator visit plus a leaf-entry visit
            const bool one_to_one = t->unfolded || t->n == 1;
            h->stats.visited = one_to_one ? pin[1] : pin[1] + total;
            h->stats.leaf_visits = one_to_one ? pin[2] : total;
        }
        if (ctx->timing) ctx->ev_set |= 4u;
        return;
    }
    throw HipFail{hipErrorUnknown, "hit pool did not converge", __LINE__};
#undef DISPATCH_WALK
}

template void traverse_batch<float>(bvhgpu_tree*, const bvhgpu_ray_f32*, size_t, unsigned, bvhgpu_hits*);
template void traverse_batch<double>(bvhgpu_tree*, const bvhgpu_ray_f64*, size_t, unsigned, bvhgpu_hits*);

// ------------------------------------------------------------------------------------------------
// <FlatBvh as BoundingHierarchy>::nearest_to (flat_bvh.rs:513-562) for a batch of query points.
// Shape distance = <Triangle as PointDistance>::distance_squared (testbase.rs:367-443: Embree's closest point on a
// triangle with degenerate-triangle guards) or the shape's own Aabb::min_distance_squared (UnitBox,
// testbase.rs:101-105; aabb_impl.rs:618-629).  Same operation order as the reference, no contraction.
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T aabb_min_dist2(const T mn[3], const T mx[3], const T p[3]) {
    T out[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const T size = mx[k] - mn[k];
        const T half = size * (T)0.5;
        const T centre = mn[k] + half;
        const T delta = p[k] - centre;
        const T q = fabs(delta) - half;
        out[k] = (q > (T)0) ? q : (T)0;   // x.max(0): NaN → 0
    }
    return dot3<T>(out, out);
}
template <typename T> __device__ __forceinline__ void closest_point_segment(const T p[3], const T a[3], const T b[3], T out[3]) {
    T ab[3], ap[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { ab[k] = b[k] - a[k]; ap[k] = p[k] - a[k]; }
    const T m = dot3<T>(ab, ab);
    T s12 = dot3<T>(ab, ap) / m;
    s12 = s12 < (T)0 ? (T)0 : (s12 > (T)1 ? (T)1 : s12);   // f32::clamp (keeps NaN)
#pragma unroll
    for (int k = 0; k < 3; k++) { const T t = s12 * ab[k]; out[k] = a[k] + t; }
}
template <typename T> __device__ void closest_point_triangle(const T p[3], const T a[3], const T b[3], const T c[3], T out[3]) {
    const bool ab_eq = a[0] == b[0] && a[1] == b[1] && a[2] == b[2];
    const bool bc_eq = b[0] == c[0] && b[1] == c[1] && b[2] == c[2];
    const bool ac_eq = a[0] == c[0] && a[1] == c[1] && a[2] == c[2];
    if (ab_eq && bc_eq && ac_eq) { out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; return; }
    if (ab_eq) { closest_point_segment<T>(p, a, c, out); return; }
    if (bc_eq || ac_eq) { closest_point_segment<T>(p, a, b, out); return; }
    T ab[3], ac[3], ap[3], bp[3], cp[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { ab[k] = b[k] - a[k]; ac[k] = c[k] - a[k]; ap[k] = p[k] - a[k]; }
    const T d1 = dot3<T>(ab, ap), d2 = dot3<T>(ac, ap);
    if (d1 <= (T)0 && d2 <= (T)0) { out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; return; }
#pragma unroll
    for (int k = 0; k < 3; k++) bp[k] = p[k] - b[k];
    const T d3 = dot3<T>(ab, bp), d4 = dot3<T>(ac, bp);
    if (d3 >= (T)0 && d4 <= d3) { out[0] = b[0]; out[1] = b[1]; out[2] = b[2]; return; }
#pragma unroll
    for (int k = 0; k < 3; k++) cp[k] = p[k] - c[k];
    const T d5 = dot3<T>(ab, cp), d6 = dot3<T>(ac, cp);
    if (d6 >= (T)0 && d5 <= d6) { out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; return; }
    const T m1 = d1 * d4, m2 = d3 * d2;
    const T vc = m1 - m2;
    if (vc <= (T)0 && d1 >= (T)0 && d3 <= (T)0) {
        const T den = d1 - d3;
        const T v = d1 / den;
#pragma unroll
        for (int k = 0; k < 3; k++) { const T t = v * ab[k]; out[k] = a[k] + t; }
        return;
    }
    const T m3 = d5 * d2, m4 = d1 * d6;
    const T vb = m3 - m4;
    if (vb <= (T)0 && d2 >= (T)0 && d6 <= (T)0) {
        const T den = d2 - d6;
        const T v = d2 / den;
#pragma unroll
        for (int k = 0; k < 3; k++) { const T t = v * ac[k]; out[k] = a[k] + t; }
        return;
    }
    const T m5 = d3 * d6, m6 = d5 * d4;
    const T va = m5 - m6;
    const T e43 = d4 - d3, e56 = d5 - d6;
    if (va <= (T)0 && e43 >= (T)0 && e56 >= (T)0) {
        const T den = e43 + e56;
        const T v = e43 / den;
#pragma unroll
        for (int k = 0; k < 3; k++) { const T cb = c[k] - b[k]; const T t = v * cb; out[k] = b[k] + t; }
        return;
    }
    T sum = va + vb;
    sum = sum + vc;
    const T denom = (T)1 / sum;
    const T v = vb * denom, w = vc * denom;
#pragma unroll
    for (int k = 0; k < 3; k++) { const T t1 = v * ab[k]; const T t2 = w * ac[k]; const T r = a[k] + t1; out[k] = r + t2; }
}
template <typename T> __device__ __forceinline__ T triangle_dist2(const T* __restrict__ tri, const T p[3]) {
    const T a[3] = {tri[0], tri[1], tri[2]}, b[3] = {tri[3], tri[4], tri[5]}, c[3] = {tri[6], tri[7], tri[8]};
    T nearest[3], diff[3];
    closest_point_triangle<T>(p, a, b, c, nearest);
#pragma unroll
    for (int k = 0; k < 3; k++) diff[k] = p[k] - nearest[k];
    return dot3<T>(diff, diff);
}

// one query point per lane, the same loop as flat_bvh.rs:533-558 over the folded array: a folded leaf entry
// stands for the navigator (min_distance_squared test of its box) followed by the leaf (exact shape distance)
template <typename T, bool TRIANGLE, bool UNFOLDED>
__global__ __launch_bounds__(256) void k_nearest(const TravNode<T>* __restrict__ nodes, uint32_t n_trav,
                                                 const T* __restrict__ shape_aabbs, const T* __restrict__ tris,
                                                 const T* __restrict__ points, uint32_t n, uint32_t* __restrict__ out_shape,
                                                 T* __restrict__ out_dist) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const T p[3] = {points[3 * (size_t)q], points[3 * (size_t)q + 1], points[3 * (size_t)q + 2]};
    bool has = false;
    T best = 0;
    uint32_t bs = NONE;
    uint32_t i = 0;
    while (i < n_trav) {
        const NodeRegs<T> nd = load_node(nodes + i);
        const bool leaf = trav_is_leaf(nd.shape);
        bool enter = true;
        if (!(UNFOLDED && leaf)) {
            const T md = aabb_min_dist2<T>(nd.mn, nd.mx, p);
            enter = !has || md < best;                           // :550
        }
        if (leaf) {
            if (enter) {
                T d;
                if (TRIANGLE) d = triangle_dist2<T>(tris + 9 * (size_t)nd.shape, p);
                else {
                    const T* sb = shape_aabbs + 6 * (size_t)nd.shape;
                    const T mn[3] = {sb[0], sb[1], sb[2]}, mx[3] = {sb[3], sb[4], sb[5]};
                    d = aabb_min_dist2<T>(mn, mx, p);
                }
                if (!has || d < best) { has = true; best = d; bs = nd.shape; }   // :540-542
            }
            i = nd.exit;
        } else {
            i = enter ? i + 1 : nd.exit;
        }
    }
    out_shape[q] = bs;
    out_dist[q] = has ? sqrt(best) : (T)0;                       // :561
}

template <typename T>
void nearest_batch(bvhgpu_tree* t, const T* points_dev, size_t n, int kind, uint32_t* out_shape_dev, T* out_dist_dev) {
    if (!n) return;
    hipStream_t st = t->ctx->stream;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const TravNode<T>* nodes = t->trav.as<TravNode<T>>();
    const uint32_t n_trav = (uint32_t)t->n_trav;
    const bool unfolded = t->unfolded || t->n == 1;   // a single-shape tree has one (leaf) entry and no navigator
#define LAUNCH_NEAREST(TRI, UNF) hipLaunchKernelGGL((k_nearest<T, TRI, UNF>), grid, block, 0, st, nodes, n_trav, t->aabbs.as<T>(), \
                                                    t->tris.as<T>(), points_dev, (uint32_t)n, out_shape_dev, out_dist_dev)
    if (kind == 1) { if (unfolded) LAUNCH_NEAREST(true, true); else LAUNCH_NEAREST(true, false); }
    else { if (unfolded) LAUNCH_NEAREST(false, true); else LAUNCH_NEAREST(false, false); }
#undef LAUNCH_NEAREST
    BVH_HIP(hipGetLastError());
}
template void nearest_batch<float>(bvhgpu_tree*, const float*, size_t, int, uint32_t*, float*);
template void nearest_batch<double>(bvhgpu_tree*, const double*, size_t, int, uint32_t*, double*);

// ------------------------------------------------------------------------------------------------
// Ray::intersects_triangle for n independent (ray, triangle) pairs — ray_impl.rs:154-213
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_ray_triangle_pairs(const typename Traits<T>::Ray* __restrict__ rays,
                                                            const T* __restrict__ tris, uint32_t n, T* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const typename Traits<T>::Ray* rp = rays + i;
    const T o[3] = {rp->o[0], rp->o[1], rp->o[2]};
    const T d[3] = {rp->d[0], rp->d[1], rp->d[2]};
    T r[3];
    ray_triangle<T>(o, d, tris + 9 * (size_t)i, r);
    out[3 * (size_t)i] = r[0]; out[3 * (size_t)i + 1] = r[1]; out[3 * (size_t)i + 2] = r[2];
}
template <typename T>
void ray_triangle_pairs(bvhgpu_ctx* ctx, const typename Traits<T>::Ray* rays_dev, const T* tris_dev, size_t n, T* out_dev) {
    if (!n) return;
    hipLaunchKernelGGL(k_ray_triangle_pairs<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, rays_dev,
                       tris_dev, (uint32_t)n, out_dev);
    BVH_HIP(hipGetLastError());
}
template void ray_triangle_pairs<float>(bvhgpu_ctx*, const bvhgpu_ray_f32*, const float*, size_t, float*);
template void ray_triangle_pairs<double>(bvhgpu_ctx*, const bvhgpu_ray_f64*, const double*, size_t, double*);

// ------------------------------------------------------------------------------------------------
// Ray::new — ray_impl.rs:70-80
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void ray_new(const T o[3], const T d[3], typename Traits<T>::Ray* out) {
    T xx = d[0] * d[0], yy = d[1] * d[1], zz = d[2] * d[2];
    T s = xx + yy;
    s = s + zz;
    T nrm = sqrt(s);  // correctly rounded (no fast-math)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        T dn = d[k] / nrm;
        out->o[k] = o[k];
        out->d[k] = dn;
        out->inv[k] = (T)1 / dn;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_rays_new(const T* __restrict__ origins, const T* __restrict__ dirs, uint32_t n,
                                                  typename Traits<T>::Ray* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T o[3] = {origins[3 * (size_t)i], origins[3 * (size_t)i + 1], origins[3 * (size_t)i + 2]};
    T d[3] = {dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2]};
    ray_new<T>(o, d, out + i);
}

template <typename T>
void rays_new(bvhgpu_ctx* ctx, const T* origins_dev, const T* dirs_dev, size_t n, typename Traits<T>::Ray* out_dev) {
    if (!n) return;
    hipLaunchKernelGGL(k_rays_new<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, origins_dev, dirs_dev,
                       (uint32_t)n, out_dev);
    BVH_HIP(hipGetLastError());
}
template void rays_new<float>(bvhgpu_ctx*, const float*, const float*, size_t, bvhgpu_ray_f32*);
template void rays_new<double>(bvhgpu_ctx*, const double*, const double*, size_t, bvhgpu_ray_f64*);

// ------------------------------------------------------------------------------------------------
// bench ray stream: create_ray (testbase.rs:687-691) over splitmix64 (:558-564), next_point3 (:567-595).
// splitmix64's state after j draws is j*GAMMA, so ray r uses states (2r+1)*GAMMA and (2r+2)*GAMMA.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ void point3_from_state(unsigned long long state, const float* bounds, float out[3]) {
    const unsigned long long u = mix64(state);
    const long long a = (long long)((u >> 32) & 0xFFFFFFFFull) - 0x80000000ll;
    const long long b = (long long)(u & 0xFFFFFFFFull) - 0x80000000ll;
    const unsigned long long ub = (unsigned long long)b;
    const unsigned long long rot = (ub << 6) | (ub >> 58);
    const long long c = a ^ (long long)rot;
    const int r[3] = {(int)a, (int)b, (int)(unsigned int)(unsigned long long)c};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float q = (float)r[k] / 2147483648.0f;  // i32::MAX as f32 == 2^31
        float fv = (q + 1.0f) * 0.5f;
        float size = bounds[3 + k] - bounds[k];
        float off = fv * size;
        out[k] = bounds[k] + off;
    }
}

struct Bounds6 { float b[6]; };

template <typename T>
__global__ __launch_bounds__(256) void k_gen_rays(unsigned long long first, uint32_t n, Bounds6 bounds,
                                                  typename Traits<T>::Ray* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long G = 0x9E3779B97F4A7C15ull;
    const unsigned long long r = first + i;
    float o[3], d[3];
    point3_from_state((2ull * r + 1ull) * G, bounds.b, o);
    point3_from_state((2ull * r + 2ull) * G, bounds.b, d);
    T oo[3] = {(T)o[0], (T)o[1], (T)o[2]};
    T dd[3] = {(T)d[0], (T)d[1], (T)d[2]};
    ray_new<T>(oo, dd, out + i);
}

// coherent primary rays (BASELINE.json configs[2]): pinhole camera, row-major W x H image.  Definition in
// include/bvh_mi355x.h (bvhgpu_gen_primary_rays_*); every operation is a separately rounded f32 op.
struct Camera14 { float c[14]; };
template <typename T>
__global__ __launch_bounds__(256) void k_gen_primary(Camera14 cam, uint32_t width, uint32_t height, unsigned long long first,
                                                     uint32_t n, typename Traits<T>::Ray* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long id = first + i;
    const uint32_t x = (uint32_t)(id % width), y = (uint32_t)(id / width);
    float fx = (float)x + 0.5f; fx = fx / (float)width; fx = fx * 2.0f; const float sx = fx - 1.0f;
    float fy = (float)y + 0.5f; fy = fy / (float)height; fy = fy * 2.0f; const float sy = 1.0f - fy;
    const float ax = sx * cam.c[12], ay = sy * cam.c[13];
    T oo[3], dd[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float r = ax * cam.c[3 + k], u = ay * cam.c[6 + k];
        const float t = cam.c[9 + k] + r;
        const float d = t + u;
        oo[k] = (T)cam.c[k];
        dd[k] = (T)d;
    }
    ray_new<T>(oo, dd, out + i);
}
template <typename T>
void gen_primary(bvhgpu_ctx* ctx, const float cam[14], uint32_t width, uint32_t height, uint64_t first, size_t n,
                 typename Traits<T>::Ray* out_dev) {
    if (!n) return;
    Camera14 c;
    for (int k = 0; k < 14; k++) c.c[k] = cam[k];
    hipLaunchKernelGGL(k_gen_primary<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, c, width, height,
                       (unsigned long long)first, (uint32_t)n, out_dev);
    BVH_HIP(hipGetLastError());
}
template void gen_primary<float>(bvhgpu_ctx*, const float*, uint32_t, uint32_t, uint64_t, size_t, bvhgpu_ray_f32*);
template void gen_primary<double>(bvhgpu_ctx*, const float*, uint32_t, uint32_t, uint64_t, size_t, bvhgpu_ray_f64*);

void gen_rays_f32(bvhgpu_ctx* ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f32* out_dev) {
    if (!n) return;
    Bounds6 b;
    for (int k = 0; k < 6; k++) b.b[k] = bounds[k];
    hipLaunchKernelGGL(k_gen_rays<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (unsigned long long)first, (uint32_t)n, b, out_dev);
    BVH_HIP(hipGetLastError());
}
void gen_rays_f64(bvhgpu_ctx* ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f64* out_dev) {
    if (!n) return;
    Bounds6 b;
    for (int k = 0; k < 6; k++) b.b[k] = bounds[k];
    hipLaunchKernelGGL(k_gen_rays<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (unsigned long long)first, (uint32_t)n, b, out_dev);
    BVH_HIP(hipGetLastError());
}

}  // namespace bvhgpu
