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
#include <cstdio>
#include <type_traits>

#include "engine.hpp"

namespace bvhgpu {

constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;   // rays per workgroup of the count scan

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
    unsigned long long* closest_key;   // closest mode with rays cut into items (f32): per ray min over its items of {key(distance) << 32 | item << 28 | shape}
                                 // (kept all-ones between batches; k_closest_resolve turns the winner into closest / closest_prim).  NULL: one lane owns the ray
    uint32_t* item_cnt;          // wide walk with several items per ray: hits of item (ray, j), written only when non-zero
    uint32_t* ray_items;         // ... and per ray the set of j that wrote one (kept all-zero between batches like counts)
    uint32_t* scan_sums;         // wide walk: hits per SCAN_BLOCK rays, added up by the workgroups as they finish (a zeroed set; NULL: k_scan_reduce does the sums)
    uint4* pool_pair;            // wide walk, whole rays, indices only: pair records {ray, k, shape, shape | NONE} (report_pair) in the pool's memory instead of the 12-byte
                                 // HitRec: 8 bytes per hit, one offset gather per two hits in the scatter.  NULL: HitRec
    uint32_t* raybuf;            // wide walk, whole rays, indices only: the first 2^stage_shift shapes of ray r go straight to raybuf[r << stage_shift | k]
    uint32_t stage_shift;        // (4 bytes per hit, no record, no atomic); only later hits of a ray become pool records.  NULL: everything through the pool
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
    // o / inv were filled by the caller (the guide walk's f32 view of an f64 ray): the rest of load()
    __device__ __forceinline__ void loaded(uint32_t ray) {
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

struct __attribute__((packed, aligned(4))) DwordPair { uint32_t a, b; };   // two consecutive CSR entries, stored at once
#ifndef SCATTER8_UNROLL
#define SCATTER8_UNROLL 16
#endif
// Whole-ray index batches (WalkOut::pool_pair) write PAIR records, 16 bytes for two hits: {ray, k of the first, shape, shape | NONE}.  A lane
// keeps one hit pending and writes a record when its ray's next hit arrives — or, alone, when the ray retires (flush) — so the pool holds
// 8 bytes per hit instead of HitRec's 12, and the scatter reads one ray offset per TWO hits: the CSR assembly of a hit-heavy batch (457 M
// hits for configs[3]'s 100 M rays) is bound by exactly those dependent gathers.  Measured against the 8-byte single-hit record
// {ray, k << 25 | shape} it replaced (profiles/r4_pair_records_*_ab.log): 12.5 M incoherent rays 2.88 -> 2.75 ms per step, 10 M primary
// rays 1.914 -> 1.868; and no limit on hits per ray or shapes per scene, so no fallback format.
__device__ __forceinline__ void pool_invalidate_tail16(uint4* pool, unsigned long long pool_cap, const PoolCursor& pc, int lane) {
    for (uint32_t j = (uint32_t)lane; j < pc.left; j += WAVE)
        if (pc.pos + j < pool_cap) pool[pc.pos + j].x = NONE;
}
template <typename RAY>
__device__ __forceinline__ void report_pair(bool rec, bool flush, uint32_t shape, RAY& ray, uint32_t& pend, uint4* pool, unsigned long long pool_cap,
                                            unsigned long long* ctr, PoolCursor& pc, int lane, unsigned long long lt) {
    const bool emit = (rec || flush) && pend != NONE;
    const unsigned long long m = __ballot(emit);
    if (m) {
        const uint32_t h = (uint32_t)__popcll(m);
        if (h > pc.left) {   // wave-uniform: start a new chunk, invalidate what is left of the old one
            pool_invalidate_tail16(pool, pool_cap, pc, lane);
            unsigned int blo = 0, bhi = 0;
            if (lane == 0) {
                unsigned long long b = atomicAdd(&ctr[0], (unsigned long long)pc.next);
                blo = (unsigned int)b; bhi = (unsigned int)(b >> 32);
            }
            blo = __builtin_amdgcn_readfirstlane(blo); bhi = __builtin_amdgcn_readfirstlane(bhi);
            pc.pos = ((unsigned long long)bhi << 32) | blo;
            pc.left = pc.next;
            pc.next = pc.next < POOL_CHUNK_MAX ? pc.next * 2 : POOL_CHUNK_MAX;
        }
        if (emit) {
            const unsigned long long slot = pc.pos + __popcll(m & lt);
            if (slot < pool_cap) pool[slot] = make_uint4(ray.r, ray.cnt - 1u, pend, rec ? shape : NONE);   // (the pending hit is number cnt - 1)
        }
        pc.pos += h; pc.left -= h;
    }
    if (rec) { pend = emit ? NONE : shape; ray.cnt++; }
    else if (emit) pend = NONE;
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

// ------------------------------------------------------------------------------------------------
// Wide walk (large incoherent batches, the default): four grandchild boxes per step instead of one child box.
//
// Why it returns the reference's list.  FlatBvh::traverse reports shape s iff the slab test passes for every
// ancestor box of s and for s's own AABB (flat_bvh.rs:408-427), in pre-order.  Every ancestor box is the exact join
// (component-wise min / max, no rounding) of the AABBs below it, so it contains s's AABB component by component.
// For a ray whose origin and inverse direction are finite, against finite boxes, no product (b - o) * inv is NaN and
// each is monotone in b (IEEE subtraction and multiplication by a constant are monotone under round-to-nearest):
// growing a box can only lower its entry parameter and raise its exit parameter, so
//        slab(ray, AABB(s)) passes  ⇒  slab(ray, every ancestor box of s) passes.
// The ancestor tests are therefore redundant for the RESULT, and a walk may skip tree levels as long as it keeps the
// pre-order: this kernel visits, for an inner node b, the four grandchildren directly (common.hpp WideNode).  On the
// 120k-triangle scene a ray needs 20 dependent steps instead of 79, for the same 79 box tests.  Rays with a non-finite
// component (axis-parallel: inv = ±inf) can produce NaN products, which the reference turns into a miss
// (intersect_default.rs:22-28) and which break the implication above; waves holding such a ray take the exact
// sequence instead: the skipped child box is rebuilt as the join of its two grandchild boxes (bit-identical to
// the builder's box up to the sign of a zero, which no product distinguishes) and tested with the reference's
// NaN-aware slab test before its grandchildren are.  Trees where a child box is NOT the join of its grandchildren
// (empty bounds after a split with no SAH winner, bvh_node.rs:225-230; uploaded FlatBvh whose shapes moved) and the
// outputs that need the reference's own visit sequence (STATS, T_SLICE) use the binary walks above.
//
// Per lane: `cur` = what to do next (an inner node, a leaf to report, or nothing) and a stack of the other hit
// grandchildren (at most 3 pushes per step; the first `stack_lds` entries per lane in LDS, entry-major, the rest in
// a global workspace; overflowing that raises a flag and the host replays the batch with the binary walk).
// The nodes with the K lowest 4-ary heap numbers (root 0, children 4q+1..4q+4) are copied to LDS by every workgroup
// (one 16-byte plane per chunk, like TopLds), a lane follows heap numbers while it is inside that set.
// A ray may be cut into 4 ITEMS, one per grandchild of the root (no ancestor test is owed, see above): item 4r+j walks
// the root with only slot j enabled; a ray's list is the concatenation of its items' lists.
// Waves are persistent and draw items from a workgroup cursor exactly like k_traverse_lds.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t CUR_NONE = 0x7FFFFFFFu;   // neither a shape index (< 2^28) nor an inner reference (bit 31)
#ifndef BVH_WIDE_INNER_STEPS
#define BVH_WIDE_INNER_STEPS 4
#endif
#ifndef BVH_WIDE_INNER_STEPS_WHOLE
#define BVH_WIDE_INNER_STEPS_WHOLE 8
#endif
#ifndef BVH_WIDE_INNER_STEPS_COHERENT
#define BVH_WIDE_INNER_STEPS_COHERENT 12
#endif
// walk steps between two refill phases: items of a ray cut into 16 are short (2 / 3 / 4 / 6 steps: 0.1225 / 0.1220 / 0.1219 / 0.1252 ms on
// configs[1]: a compile-time 4), whole rays walk for hundreds of steps (a kernel argument: 4 / 6 / 8 / 12 / 16 → 1.50 / 1.45 / 1.42 / 1.38 / 1.41 ms for
// 10 M primary rays on the stand-in scene, 2.28 / 2.28 / 2.26 / 2.27 / 2.30 for a 12.5 M-ray incoherent shard: 12 for batches the caller
// calls COHERENT, 8 otherwise)
#ifndef BVH_WIDE_MIN_WAVES_F32
#define BVH_WIDE_MIN_WAVES_F32 8   // __launch_bounds__: waves per SIMD the f32 kernel must allow (8 = two 1024-thread workgroups per CU)
#endif
#ifndef BVH_WIDE_MIN_WAVES_F64
#define BVH_WIDE_MIN_WAVES_F64 4   // f64: two 512-thread workgroups per CU
#endif
// An ITEM is (ray, j): the part of a ray's walk below the j-th of the 4^L subtrees L wide levels under the root (L = 1: the
// root's grandchildren, L = 2: their grandchildren).  No ancestor test is owed for a finite ray (see above), so items
// are independent walks and a ray's list is the concatenation of its items' lists in j order.  A workgroup tests each of
// its rays against the 4^L subtree boxes first and keeps the items whose box is hit in a compact list (62 % / 86 % of
// the items of the BASELINE stream die there); the walk then only ever draws live items.  Why: at 1 M rays a resident
// lane gets two rays, and the launch lasts as long as its unluckiest lanes (up to 66 dependent steps per ray); items of
// a quarter / a sixteenth of that length pack the lanes better (simulated critical path per workgroup 80 → 63 → 54 steps).
// Rays with a non-finite component are not cut: they travel as one item (j = WIDE_ITEM_WHOLE) from the root.
constexpr uint32_t WIDE_ITEM_BITS = 5;                 // item = ray << 5 | j
constexpr uint32_t WIDE_ITEM_WHOLE = 16;               // j of an uncut ray (its hits are filed under j = 0)
constexpr uint32_t WIDE_BSUM_MAX = 128;                // 64-ray blocks per workgroup up to which the walk keeps the scan's block sums
constexpr size_t WIDE_ITEM_MAX_RAYS = (size_t)1 << 27;

template <typename T> struct WideRegs { T mn[3][4], mx[3][4]; uint32_t ref[4]; };
template <typename T> struct WideIo {
    static constexpr int CHUNKS = (int)(sizeof(WideRegs<T>) / 16);   // 7 (f32) / 13 (f64) 16-byte chunks per node
    static_assert(sizeof(WideRegs<T>) % 16 == 0, "wide regs");
    static __device__ __forceinline__ WideRegs<T> from_global(const WideNode<T>* __restrict__ p) {
        const uint4* q = reinterpret_cast<const uint4*>(p);
        uint4 c[CHUNKS];
#pragma unroll
        for (int j = 0; j < CHUNKS; j++) c[j] = q[j];
        WideRegs<T> r;
        __builtin_memcpy(&r, c, sizeof r);
        return r;
    }
    // LDS copy: node-major, CHUNKS x 16 bytes per slot, so the chunk offsets are immediates of the ds_read_b128s
    static __device__ __forceinline__ WideRegs<T> from_lds(const uint4* nodes, uint32_t slot) {
        const uint4* q = nodes + __umul24(slot, (uint32_t)CHUNKS);   // (24-bit multiply: full rate, v_mul_lo_u32 is quarter rate)
        uint4 c[CHUNKS];
#pragma unroll
        for (int j = 0; j < CHUNKS; j++) c[j] = q[j];
        WideRegs<T> r;
        __builtin_memcpy(&r, c, sizeof r);
        return r;
    }
};

// reference of the subtree in slot `c` of a node that sits in LDS slot `q`: grandchildren that are resident too are named by
// their LDS slot (4-ary heap number), so that the walk never has to translate
__device__ __forceinline__ uint32_t wide_resident_ref(uint32_t ref, uint32_t q, uint32_t c, uint32_t K) {
    const uint32_t cs = 4u * q + 1u + c;
    return (ref != NONE && (ref & WIDE_INNER) && cs < K) ? (WIDE_INNER | WIDE_RESIDENT | cs) : ref;
}

// the four slab tests of one wide node → hit bits.  EXACT: the reference's NaN-aware sequence with the skipped child
// boxes rebuilt and tested first (see the header above).
template <typename T, bool EXACT>
__device__ __forceinline__ uint32_t wide_hits(const T o[3], const T inv[3], const WideRegs<T>& nd) {
    uint32_t m = 0;
    if (!EXACT) {   // absent slots carry NaN boxes: v_min / v_max3 keep the NaN and both compares fail
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const T mn[3] = {nd.mn[0][c], nd.mn[1][c], nd.mn[2][c]}, mx[3] = {nd.mx[0][c], nd.mx[1][c], nd.mx[2][c]};
            m |= slab_hit_finite<T>(o, inv, mn, mx) ? (1u << c) : 0u;
        }
        return m;
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int c0 = 2 * p, c1 = 2 * p + 1;
        const T mn0[3] = {nd.mn[0][c0], nd.mn[1][c0], nd.mn[2][c0]}, mx0[3] = {nd.mx[0][c0], nd.mx[1][c0], nd.mx[2][c0]};
        T t0, t1;
        if (nd.ref[c1] == NONE) {   // the child is a leaf (or absent): its own box is in slot c0
            if (nd.ref[c0] != NONE && slab_hit<T>(o, inv, mn0, mx0, t0, t1)) m |= 1u << c0;
        } else {
            const T mn1[3] = {nd.mn[0][c1], nd.mn[1][c1], nd.mn[2][c1]}, mx1[3] = {nd.mx[0][c1], nd.mx[1][c1], nd.mx[2][c1]};
            T jmn[3], jmx[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { jmn[k] = tmin(mn0[k], mn1[k]); jmx[k] = tmax(mx0[k], mx1[k]); }
            if (slab_hit<T>(o, inv, jmn, jmx, t0, t1)) {
                if (slab_hit<T>(o, inv, mn0, mx0, t0, t1)) m |= 1u << c0;
                if (slab_hit<T>(o, inv, mn1, mx1, t0, t1)) m |= 1u << c1;
            }
        }
    }
    return m;
}

// The 4^L item subtrees of a tree: box + reference, in pre-order (j = 4 * slot at wide level 1 + slot at wide level 2).  Every
// workgroup that needs them derives them itself from the root's wide node (and its four children's): two dependent loads.
#ifndef BVH_WIDE_LONG_FRAC
#define BVH_WIDE_LONG_FRAC 0.2   // of the box diagonal; measured on configs[1]: none 127 / 0.1 125 / 0.2 121 / 0.3 123.5 / 0.5 126.5 us
#endif
template <typename T> struct ItemTable {
    T box[16][6];
    uint32_t ref[16];
    T half_diag[16];   // scheduling only: an item whose ray stays inside the box for more than this is walked early (long walk expected)
};
template <typename T, int ITEMS_LOG4>
__device__ __forceinline__ void item_table_build(const WideNode<T>* __restrict__ wide, ItemTable<T>* tb, uint32_t tid) {
    constexpr uint32_t ITEMS = 1u << (2 * ITEMS_LOG4);
    if (tid < ITEMS) {
        const uint32_t c = ITEMS_LOG4 == 2 ? tid >> 2 : tid, k = tid & 3u;
        const WideNode<T>* root = wide;   // tree node 0
        uint32_t ref = root->ref[c];
        T b[6];
#pragma unroll
        for (int a = 0; a < 3; a++) { b[a] = root->mn[a][c]; b[3 + a] = root->mx[a][c]; }
        if (ITEMS_LOG4 == 2) {
            if (ref != NONE && (ref & WIDE_INNER)) {   // an inner grandchild of the root: its own four grandchildren
                const WideNode<T>* g = wide + (ref & (WIDE_RESIDENT - 1u));
                ref = g->ref[k];
#pragma unroll
                for (int a = 0; a < 3; a++) { b[a] = g->mn[a][k]; b[3 + a] = g->mx[a][k]; }
            } else if (k != 0) {                       // a leaf (or nothing): the whole of it is item 4c
                ref = NONE;
            }
        }
        if (ref == NONE) {
            const T nan = __builtin_nan("");
#pragma unroll
            for (int a = 0; a < 6; a++) b[a] = nan;
        }
#pragma unroll
        for (int a = 0; a < 6; a++) tb->box[tid][a] = b[a];
        tb->ref[tid] = ref;
        const T dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
        tb->half_diag[tid] = (T)BVH_WIDE_LONG_FRAC * sqrt(dx * dx + dy * dy + dz * dz);
    }
}
// LDS slot (4-ary heap number) of item j's subtree root
template <int ITEMS_LOG4> __device__ __forceinline__ uint32_t item_slot(uint32_t j) {
    return ITEMS_LOG4 == 2 ? 5u + j : 1u + j;   // level 1: 1 + c; level 2: 4 * (1 + c) + 1 + k = 5 + 4c + k
}

// The rays of one workgroup of the wide walk (64-ray blocks b, b + G, b + 2G, ... of the batch) → its live items, written
// into its own region of the list: the 4^L subtree boxes are tested, the survivors compacted per wave (one LDS atomic per
// wave and list end).  Items whose ray stays long inside their subtree's box (long walks expected) fill the list from the
// front, the others from the back — the walk draws from the front, so that the longest chains start first instead of setting
// the end of the launch.  Called by the walk's prologue or, earlier and beside the build, by k_wide_items.
struct GuideArgs;
__device__ __forceinline__ void guide_ray_load(const bvhgpu_ray_f64* __restrict__ rays64, uint32_t r, double S, float o[3], float inv[3], bool& bad);
// rays64 != NULL (GUIDE, T = float): the batch is an f64 one — every ray is converted where it is loaded (guide_ray_load) and *any_bad
// collects whether one of them lies outside the guide walk's range
template <typename T, int L4, bool GUIDE = false>
__device__ __forceinline__ void filter_rays_into_list(const ItemTable<T>* tb, const typename Traits<T>::Ray* __restrict__ rays, uint32_t n_rays,
                                                      uint32_t* __restrict__ list, uint32_t per_wg, uint32_t my_rays, uint32_t G, uint32_t b,
                                                      uint32_t tid, uint32_t bd, int lane, uint32_t* s_nlist, uint32_t* s_nback,
                                                      const bvhgpu_ray_f64* __restrict__ rays64 = nullptr, double guide_S = 0.0, bool* any_bad = nullptr) {
    constexpr uint32_t ITEMS = 1u << (2 * L4);
    for (uint32_t l0 = 0; l0 < my_rays; l0 += bd) {   // workgroup-uniform
        const uint32_t local = l0 + tid;
        const uint32_t r = local < my_rays ? ((((local >> 6) * G + b) << 6) | (local & 63u)) : n_rays;
        uint32_t mask = 0, longm = 0;
        if (r < n_rays) {
            T o[3], inv[3];
            if constexpr (GUIDE) {
                bool bad;
                guide_ray_load(rays64, r, guide_S, o, inv, bad);
                *any_bad = *any_bad || bad;
            } else {
                const typename Traits<T>::Ray* rp = rays + r;
#pragma unroll
                for (int k = 0; k < 3; k++) { o[k] = rp->o[k]; inv[k] = rp->inv[k]; }
            }
            if (!ray_is_finite<T>(o, inv)) {
                mask = 1u << WIDE_ITEM_WHOLE; longm = mask;
            } else {
#pragma unroll 4
                for (uint32_t j = 0; j < ITEMS; j++) {   // the boxes are workgroup-uniform: LDS broadcast reads
                    const T mn[3] = {tb->box[j][0], tb->box[j][1], tb->box[j][2]}, mx[3] = {tb->box[j][3], tb->box[j][4], tb->box[j][5]};
                    T len;
                    const bool hit = slab_hit_finite_len<T>(o, inv, mn, mx, len);
                    mask |= hit ? (1u << j) : 0u;
                    longm |= (hit && len > tb->half_diag[j]) ? (1u << j) : 0u;
                }
            }
        }
        const uint32_t cap = per_wg * ITEMS;
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const uint32_t mm0 = side ? (mask & ~longm) : (mask & longm);
            const uint32_t mine = (uint32_t)__popc(mm0);
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const uint32_t u = __shfl_up(incl, d);
                if (lane >= d) incl += u;
            }
            const uint32_t total = __shfl(incl, WAVE - 1);
            uint32_t base = 0;
            if (lane == 0 && total) base = atomicAdd(side ? s_nback : s_nlist, total);
            base = __shfl(base, 0) + incl - mine;
            uint32_t mm = mm0;
            while (mm) {
                const uint32_t bit = (uint32_t)__ffs(mm) - 1u;
                mm &= mm - 1u;
                list[side ? cap - 1u - base : base] = (r << WIDE_ITEM_BITS) | bit;
                base++;
            }
        }
    }
}

// The item filter of a batch, EARLY: launched on the ctx's side stream behind the level pass that splits tree level 3, it
// runs beside the rest of the build (nine tenths of which leave the chip idle) instead of in front of the walk — the walk's
// prologue shrinks from 17-28 µs to the LDS image load.  The 16 item boxes are those of tree level 4 (heap numbers 16..31),
// read out of the BvhNode records of levels 0..3, which are final by then IF the level tier wrote them all (counter
// CTR_TOPMASK) and every level-4 node is an inner node; otherwise the kernel says so (front = NONE) and every workgroup of the
// walk filters its rays itself, as it does for a tree that is not being rebuilt.  Same grid as the walk (workgroup b owns the
// same 64-ray blocks and the same list region), a quarter of its threads: it is a guest on the chip.
template <typename T>
__global__ __launch_bounds__(256) void k_wide_items(const typename Traits<T>::Node* __restrict__ nodes, uint32_t n_nodes,
                                                    const uint32_t* __restrict__ node_count, const uint32_t* __restrict__ build_ctr,
                                                    const typename Traits<T>::Ray* __restrict__ rays, uint32_t n_rays,
                                                    uint32_t* __restrict__ list_all, uint32_t* __restrict__ wg_items) {
    constexpr int L4 = 2;
    constexpr uint32_t ITEMS = 16;
    __shared__ ItemTable<T> tb;
    __shared__ uint32_t s_nlist, s_nback, s_bad;
    const uint32_t tid = threadIdx.x, bd = blockDim.x;
    const int lane = lane_id();
    if (tid == 0) { s_nlist = 0u; s_nback = 0u; s_bad = (build_ctr[BUILD_CTR_TOPMASK] & 0xFFFEu) == 0xFFFEu ? 0u : 1u; }
    __syncthreads();
    if (s_bad) { if (tid == 0) wg_items[2u * blockIdx.x] = NONE; return; }
    if (tid < ITEMS) {   // item j = subtree of heap number 16 + j: four steps down from the root, its box is in its parent's record
        uint32_t node = 0;
        T bx[6];
        bool ok = n_nodes > 1u;
        uint32_t cnt = ok ? node_count[0] : 0u;
#pragma unroll
        for (int lv = 3; lv >= 0 && ok; lv--) {
            const typename Traits<T>::Node nd = nodes[node];
            const uint32_t right = ((16u + tid) >> lv) & 1u;
            ok = nd.shape == NONE && nd.l < n_nodes && nd.r < n_nodes && nd.r > nd.l;
            if (!ok) break;
            const uint32_t nl = (nd.r - nd.l + 1u) >> 1;   // the left subtree holds 2 nl - 1 nodes (bvh_node.rs:138-142)
            cnt = right ? cnt - nl : nl;
            node = right ? nd.r : nd.l;
#pragma unroll
            for (int k = 0; k < 3; k++) { bx[k] = right ? nd.r_min[k] : nd.l_min[k]; bx[3 + k] = right ? nd.r_max[k] : nd.l_max[k]; }
        }
        ok = ok && cnt > 1u;   // the item's root must be an inner node (the walk names it by its wide node)
        if (!ok) atomicOr(&s_bad, 1u);
#pragma unroll
        for (int k = 0; k < 6; k++) tb.box[tid][k] = bx[k];
        const T dx = bx[3] - bx[0], dy = bx[4] - bx[1], dz = bx[5] - bx[2];
        tb.half_diag[tid] = (T)BVH_WIDE_LONG_FRAC * sqrt(dx * dx + dy * dy + dz * dz);
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) wg_items[2u * blockIdx.x] = NONE; return; }
    const uint32_t n_blocks = (n_rays + 63u) >> 6;
    const uint32_t my_blocks = n_blocks > blockIdx.x ? (n_blocks - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0u;
    const uint32_t per_wg = ((n_blocks + gridDim.x - 1u) / gridDim.x) << 6;
    uint32_t* list = list_all + (size_t)blockIdx.x * per_wg * ITEMS;
    filter_rays_into_list<T, L4>(&tb, rays, n_rays, list, per_wg, my_blocks << 6, gridDim.x, blockIdx.x, tid, bd, lane, &s_nlist, &s_nback);
    __syncthreads();
    if (tid == 0) { wg_items[2u * blockIdx.x] = s_nlist; wg_items[2u * blockIdx.x + 1u] = s_nback; }
}

// ---- guide walk: an f64 index batch walked over the tree's f32 guide boxes (common.hpp "guide boxes") -----------------------------------
// The f64 wide walk costs 1.9 x the f32 one (half-rate VALU, 13 instead of 7 chunks per node).  Only leaf tests decide a ray's list
// (monotonicity), so every inner test may be conservative: the walk converts every f64 ray to f32 (round to nearest) where it loads it (guide_ray_load) and
// flags rays the containment argument does not cover; the f32 wide walk then runs over `wide_guide` unchanged, except that a leaf
// CANDIDATE is confirmed by the f64 slab test of the shape's own f64 box with the f64 ray before it is reported.  Same lists, same order.
constexpr unsigned long long WALK_FLAG_GUIDE_RANGE = 16ull;   // ctr[7] bit: a ray was outside the guide walk's range — the host replays in f64
struct GuideArgs { const bvhgpu_ray_f64* rays64; const double* aabbs64; const float* info; const double* tris64; };   // info[0] = S (tree->guide_info); tris64: closest-hit batches
// The guide walk's f32 view of an f64 ray, made where the ray is loaded (round 4: a kernel of its own wrote an f32 copy of the batch first
// — 72 B read + 36 B written per ray and a launch, 19-22 µs per 1 M rays): origin and 1/d rounded to nearest, and the range test of the
// containment argument (common.hpp "guide boxes"); `bad` = the argument does not cover this ray.
__device__ __forceinline__ void guide_ray_load(const bvhgpu_ray_f64* __restrict__ rays64, uint32_t r, double S, float o[3], float inv[3], bool& bad) {
    const bvhgpu_ray_f64* q = rays64 + r;
    bad = !(S >= GUIDE_SCENE_MIN) || !(S <= GUIDE_SCENE_MAX);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double ok = q->o[k], ik = q->inv[k];
        const double ao = fabs(ok), ainv = fabs(ik), ai = ainv * (4.0 * S);
        // (NaN fails every comparison; S = 0 — a scene that is one point — leaves no room for the growth)
        bad = bad || !(ao <= GUIDE_ORIGIN_MAX * S) || !(ao <= GUIDE_F32_MAX) || !(ai <= 0x1p100) || !(ai >= 0x1p-100) ||
              !(ainv <= GUIDE_F32_MAX) || !(ainv >= GUIDE_F32_MIN_NORMAL);
        o[k] = (float)ok; inv[k] = (float)ik;
    }
}
// the f64 test of a leaf candidate (finite ray: the NaN-free form is exact, common.hpp slab_hit_finite)
__device__ __forceinline__ bool guide_leaf_hit(const GuideArgs& ga, uint32_t ray, uint32_t shape) {
    const bvhgpu_ray_f64* rp = ga.rays64 + ray;
    const double* b = ga.aabbs64 + 6 * (size_t)shape;
    const double o[3] = {rp->o[0], rp->o[1], rp->o[2]}, inv[3] = {rp->inv[0], rp->inv[1], rp->inv[2]};
    const double mn[3] = {b[0], b[1], b[2]}, mx[3] = {b[3], b[4], b[5]};
    return slab_hit_finite<double>(o, inv, mn, mx);
}

// a leaf candidate of the guide walk that passed its f64 box, in a closest-hit batch: Ray::intersects_triangle in f64 (ray_impl.rs:154-213) and the
// reference's strict < against the lane's nearest so far (testbase.rs:831-833).  Inlined, although candidates are rare and the f64
// Möller–Trumbore needs more registers than the f32 walk around it owns: as a real call (__noinline__) the walk took 0.208 ms instead of
// 0.141 — the calling convention's register split costs the hot loop more than the spills around the rare branch do.
#ifndef BVH_GUIDE_CANDIDATE_ATTR
#define BVH_GUIDE_CANDIDATE_ATTR __forceinline__
#endif
__device__ BVH_GUIDE_CANDIDATE_ATTR double guide_candidate_distance(const bvhgpu_ray_f64* __restrict__ rays64, const double* __restrict__ tris64, uint32_t ray, uint32_t shape) {
    const bvhgpu_ray_f64* rp = rays64 + ray;
    const double o[3] = {rp->o[0], rp->o[1], rp->o[2]}, d[3] = {rp->d[0], rp->d[1], rp->d[2]};
    double out[3];
    ray_triangle<double>(o, d, tris64 + 9 * (size_t)shape, out);
    return out[0];
}

#ifdef BVH_WIDE_PROFILE   // developer build: per-wave timestamps (100 MHz wall clock) of the wide walk's phases
__device__ unsigned long long g_wide_prof[4 * 16384];
// lane-utilisation counts per wave (16 per wave): [0] wave-steps, [1] lanes on an inner node, [2] steps with a resident fetch,
// [3] steps with a non-resident fetch, [4] steps with a lane on the slow push path, [5] lanes on it, [6] lanes reporting a leaf,
// [7] steps with a report, [8] lanes holding an item (x steps), [9] refill rounds, [10] rounds on the exact (non-finite) path,
// [11] steps with a pop from the HBM part of the stack, [12] boxes hit (sum of popc(m)), [13] lanes whose node had no hit
__device__ unsigned long long g_wide_util[16 * 16384];
#endif
// ITEMS_LOG4 = 0: one item per ray, drawn by ray number.  1 / 2: every workgroup first cuts ITS rays into live items (its
// region of `list`, filled through an LDS counter — no global atomic: one address only takes ~88 atomics per µs on this
// chip, which made a separate filter kernel with one atomic per wave cost more than the walk) and then walks them.
template <typename T, int MODE, int ITEMS_LOG4, int MAX_THREADS, int MIN_WAVES, int GUIDE = 0>
__global__ __launch_bounds__(MAX_THREADS, MIN_WAVES) void k_traverse_wide(
    const WideNode<T>* __restrict__ wide, const uint32_t* __restrict__ wslot_node, uint32_t K, uint32_t stack_lds,
    const typename Traits<T>::Ray* __restrict__ rays, uint32_t n_rays, uint32_t* __restrict__ list_all, const uint32_t* __restrict__ wg_items,
    WalkOut<T> w, uint32_t* __restrict__ gstack, uint32_t gstack_cap, uint32_t* __restrict__ overflow, GuideArgs ga, uint32_t whole_steps) {
    static_assert(GUIDE == 0 || ((MODE == MODE_INDICES || MODE == MODE_CLOSEST) && sizeof(T) == 4), "the guide walk is the f32 walk of an f64 index / closest-hit batch");
    constexpr bool GUIDE_CLOSEST = GUIDE != 0 && MODE == MODE_CLOSEST;   // candidates are decided in f64 (guide_closest_candidate); the lane keeps (distance, shape)
    double gbest = 0.0;
    static_assert(ITEMS_LOG4 >= 0 && ITEMS_LOG4 <= 2, "1, 4 or 16 items per ray");
    static_assert(MODE != MODE_T_SLICE, "the t-slice output walks the binary array");
    constexpr int CH = WideIo<T>::CHUNKS;
    constexpr int L4 = ITEMS_LOG4 > 0 ? ITEMS_LOG4 : 1;      // (so that the item code compiles when it is not used)
    constexpr uint32_t ITEMS = 1u << (2 * L4);
    extern __shared__ __attribute__((aligned(16))) uint4 wsmem[];
    uint32_t& s_next = *reinterpret_cast<uint32_t*>(wsmem);
    uint32_t& s_nlist = *(reinterpret_cast<uint32_t*>(wsmem) + 1);
    uint32_t& s_nback = *(reinterpret_cast<uint32_t*>(wsmem) + 2);
    uint32_t* s_item_ref = reinterpret_cast<uint32_t*>(wsmem + 1);           // 16 references (64 bytes)
    uint4* nodes = wsmem + 5;
    uint32_t* s_stack = reinterpret_cast<uint32_t*>(nodes + (size_t)CH * K);
    const uint32_t bd = blockDim.x, tid = threadIdx.x;
    const int WIDE_INNER_STEPS = ITEMS_LOG4 == 0 ? (int)whole_steps : BVH_WIDE_INNER_STEPS;
    constexpr uint32_t SB = MAX_THREADS;   // stride of the LDS stack's entry planes: a constant, so that the three stores of a push share one address register
    const size_t G = (size_t)gridDim.x * bd, gid = (size_t)blockIdx.x * bd + tid;
    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    // GUIDE: `rays` is unused — the batch is ga.rays64, every ray converted to f32 where it is loaded; guide_bad = one of this lane's
    // rays was outside the range the containment argument covers (the wave raises WALK_FLAG_GUIDE_RANGE at the end: the host replays in f64)
    double guide_S = 0.0;
    bool guide_bad = false;
    if constexpr (GUIDE != 0) guide_S = (double)ga.info[0];
#ifdef BVH_WIDE_PROFILE
    const unsigned long long prof_t0 = wall_clock64();
    unsigned long long prof_steps = 0;
    unsigned long long pu[16];
    for (int i = 0; i < 16; i++) pu[i] = 0;
    uint32_t pl_boxes = 0, pl_nohit = 0;
#endif
    // This workgroup's rays: the 64-ray blocks b, b + grid, b + 2 grid, ... of the batch.  (Contiguous ranges per workgroup
    // put all of a stream's expensive stretch — the BASELINE stream's first 5 000 rays start inside a cube — on a few
    // workgroups: the slowest workgroup finished at 158 µs against a mean of 115 µs.)
    const uint32_t n_blocks = (n_rays + 63u) >> 6;
    const uint32_t my_blocks = n_blocks > blockIdx.x ? (n_blocks - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0u;
    const uint32_t per_wg = ((n_blocks + gridDim.x - 1u) / gridDim.x) << 6;   // capacity of a workgroup's share (host: the same formula)
    const uint32_t my_rays = my_blocks << 6;                                   // local ray numbers [0, my_rays), some beyond n_rays in the last block
    auto ray_of = [&](uint32_t local) -> uint32_t { return (((local >> 6) * gridDim.x + blockIdx.x) << 6) | (local & 63u); };
    // hits per 64-ray block of this workgroup (local block numbers): retiring items add to them, the workgroup stores them at
    // the end — the CSR scan then needs no reduce pass over the counts.  (LDS atomics: adding straight into global sums put
    // the BASELINE stream's 10 000 hits on one cache line, 128 → 162 µs.)
    __shared__ uint32_t s_bsum[WIDE_BSUM_MAX];
    if (w.scan_sums) for (uint32_t b = tid; b < WIDE_BSUM_MAX; b += bd) s_bsum[b] = 0u;
    if (tid == 0) { s_next = 0u; s_nlist = 0u; s_nback = 0u; }
    for (uint32_t q = tid; q < K; q += bd) {
        const uint32_t node = wslot_node[q];
        if (node != NONE) {
            const uint4* src = reinterpret_cast<const uint4*>(wide + node);
            uint4* dst = nodes + (size_t)q * CH;
#pragma unroll
            for (int c = 0; c < CH - 1; c++) dst[c] = src[c];
            uint4 rf = src[CH - 1];   // the four references: resident grandchildren by LDS slot
            rf.x = wide_resident_ref(rf.x, q, 0u, K); rf.y = wide_resident_ref(rf.y, q, 1u, K);
            rf.z = wide_resident_ref(rf.z, q, 2u, K); rf.w = wide_resident_ref(rf.w, q, 3u, K);
            dst[CH - 1] = rf;
        }
    }
    uint32_t* list = nullptr;
    if (ITEMS_LOG4 > 0) {
        __shared__ ItemTable<T> tb;
        item_table_build<T, L4>(wide, &tb, tid);
        __syncthreads();
        if (tid < ITEMS) {   // references of the item subtrees, resident ones by LDS slot
            const uint32_t ref = tb.ref[tid], slot = item_slot<L4>(tid);
            s_item_ref[tid] = (ref != NONE && (ref & WIDE_INNER) && slot < K) ? (WIDE_INNER | WIDE_RESIDENT | slot) : ref;
        }
        // rays → live items, into this workgroup's region of the list (at most ITEMS per ray) — unless the batch's early filter
        // (k_wide_items, enqueued beside the build of the tree) has done it already
        list = list_all + (size_t)blockIdx.x * per_wg * ITEMS;
        const uint32_t pre_front = wg_items ? wg_items[2u * blockIdx.x] : NONE;   // workgroup-uniform
        if (pre_front != NONE) {
            if (tid == 0) { s_nlist = pre_front; s_nback = wg_items[2u * blockIdx.x + 1u]; }
        } else {
            if constexpr (GUIDE != 0) filter_rays_into_list<T, L4, true>(&tb, rays, n_rays, list, per_wg, my_rays, gridDim.x, blockIdx.x, tid, bd, lane, &s_nlist, &s_nback,
                                                                            ga.rays64, guide_S, &guide_bad);
            else filter_rays_into_list<T, L4>(&tb, rays, n_rays, list, per_wg, my_rays, gridDim.x, blockIdx.x, tid, bd, lane, &s_nlist, &s_nback);
        }
        __threadfence_block();
    }
    __syncthreads();
    const uint32_t wg_begin = 0u;
    const uint32_t n_front = s_nlist;
    const uint32_t wg_end = ITEMS_LOG4 == 0 ? my_rays : n_front + s_nback;
#ifdef BVH_WIDE_PROFILE
    const unsigned long long prof_t1 = wall_clock64();
#endif

    LaneRay<T, MODE> ray;
    ray.clear();
    uint32_t cur = CUR_NONE, sp = 0, item = NONE;
    bool exhausted = wg_begin >= wg_end;   // wave-uniform: the workgroup's range has been handed out
    bool ovf = false;
    uint32_t pair_pend = NONE;   // pair records: the lane's hit that waits for its ray's next one
    PoolCursor pc;
    auto push_slow = [&](uint32_t v) {
        if (sp < stack_lds) s_stack[sp * SB + tid] = v;
        else if (sp - stack_lds < gstack_cap) gstack[(size_t)(sp - stack_lds) * G + gid] = v;
        else ovf = true;
        sp++;
    };
    auto pop_or_none = [&]() -> uint32_t {
        if (sp == 0) return CUR_NONE;
        sp--;
        if (sp < stack_lds) return s_stack[sp * SB + tid];
        return sp - stack_lds < gstack_cap ? gstack[(size_t)(sp - stack_lds) * G + gid] : CUR_NONE;
    };
    while (true) {
        // ---- refill phase
        bool run = cur != CUR_NONE;
        const unsigned long long idle = __ballot(!run);
        if (idle) {
            if (MODE == MODE_INDICES && ITEMS_LOG4 == 0 && w.pool_pair)   // (wave-uniform) a retiring ray's unpaired last hit
                report_pair(false, !run && item != NONE, 0u, ray, pair_pend, w.pool_pair, w.pool_cap, w.ctr, pc, lane, lt);
            if (!run && item != NONE) {   // the item has left the tree: its part of the ray's list is complete
                if (MODE == MODE_CLOSEST) {
                    if constexpr (ITEMS_LOG4 == 0) {
                        const size_t r = item;
                        if constexpr (!GUIDE_CLOSEST) { w.closest[3 * r] = ray.best[0]; w.closest[3 * r + 1] = ray.best[1]; w.closest[3 * r + 2] = ray.best[2]; }
                        w.closest_prim[r] = ray.best_prim;   // (guide: the shape only — k_closest_from_prim recomputes its Intersection in f64)
                    } else if (ray.best_prim != NONE) {
                        // The ray's other items sit in other lanes: the nearest candidate of the RAY is the minimum over its items of (distance, item
                        // number) — items are the tree-level-4 subtrees in pre-order, so on equal distances the lower item holds the candidate the
                        // reference's loop meets first (strict <, testbase.rs:831-833 behind flat_bvh.rs:408), and inside an item this lane kept the
                        // first one.  Distance (monotone key), item and shape (< 2^28: WIDE_MAX_SHAPES) fit one 64-bit word: one atomicMin.
                        const uint32_t j = item & ((1u << WIDE_ITEM_BITS) - 1u);
                        const uint32_t jj = j == WIDE_ITEM_WHOLE ? 0u : j;
                        if constexpr (sizeof(T) == 4 && !GUIDE_CLOSEST) {
                            const unsigned long long key = ((unsigned long long)Traits<T>::key(ray.best[0]) << 32) | ((unsigned long long)jj << 28) | (unsigned long long)ray.best_prim;
                            atomicMin(&w.closest_key[item >> WIDE_ITEM_BITS], key);
                        } else {
                            // f64: a 64-bit distance leaves no room for item and shape in one word.  Every item that found a candidate files its shape
                            // under (ray, item) — the slots the index walk uses for hit counts — and marks itself in the ray's item set;
                            // k_closest_resolve_slots walks the set in item order with the reference's strict <, recomputing each candidate's
                            // Intersection (the same instruction sequence: the same bits)
                            const size_t r = item >> WIDE_ITEM_BITS;
                            w.item_cnt[(r << (2 * ITEMS_LOG4)) + jj] = ray.best_prim;
                            atomicOr(&w.ray_items[r], 1u << jj);
                        }
                    }
                } else if (ray.cnt) {
                    uint32_t r = item;
                    if (ITEMS_LOG4 == 0) {
                        w.counts[item] = ray.cnt;
                    } else {
                        const uint32_t j = item & ((1u << WIDE_ITEM_BITS) - 1u);
                        const uint32_t jj = j == WIDE_ITEM_WHOLE ? 0u : j;
                        r = item >> WIDE_ITEM_BITS;
                        atomicAdd(&w.counts[r], ray.cnt);
                        atomicOr(&w.ray_items[r], 1u << jj);
                        w.item_cnt[((size_t)r << (2 * ITEMS_LOG4)) + jj] = ray.cnt;
                    }
                    if (w.scan_sums) atomicAdd(&s_bsum[((r >> 6) - blockIdx.x) / gridDim.x], ray.cnt);
                }
                item = NONE;
            }
            if (!exhausted) {
                const uint32_t nidle = (uint32_t)__popcll(idle);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_next, nidle);
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t mine = base + (uint32_t)__popcll(idle & lt);
                if (!run && base < wg_end && mine < wg_end) {
                    if (ITEMS_LOG4 == 0) {
                        item = ray_of(mine);
                        if (item < n_rays) {
                            if constexpr (GUIDE != 0) { bool bad; guide_ray_load(ga.rays64, item, guide_S, ray.o, ray.inv, bad); guide_bad = guide_bad || bad; ray.loaded(item); gbest = __builtin_inf(); }
                            else ray.load(rays, item);
                            cur = WIDE_INNER | WIDE_RESIDENT | 0u;   // the root is heap slot 0 (K >= 1)
                        } else {
                            item = NONE;                         // padding of the batch's last 64-ray block
                        }
                    } else {
                        item = list[mine < n_front ? mine : per_wg * ITEMS - 1u - (mine - n_front)];
                        const uint32_t j = item & ((1u << WIDE_ITEM_BITS) - 1u);
                        if constexpr (GUIDE != 0) { bool bad; guide_ray_load(ga.rays64, item >> WIDE_ITEM_BITS, guide_S, ray.o, ray.inv, bad); ray.loaded(item >> WIDE_ITEM_BITS); gbest = __builtin_inf(); }   // (the filter has looked at its range)
                        else ray.load(rays, item >> WIDE_ITEM_BITS);
                        cur = j == WIDE_ITEM_WHOLE ? (WIDE_INNER | WIDE_RESIDENT | 0u) : s_item_ref[j];
                        if (j == WIDE_ITEM_WHOLE) item = item & ~((1u << WIDE_ITEM_BITS) - 1u);   // filed under j = 0
                    }
                    ray.r = item;                                // pool records are per item
                    sp = 0;
                    run = cur != CUR_NONE;
                }
                exhausted = base >= wg_end || (wg_end - base) <= nidle;
            }
            if (!__any(run)) break;
        }
        const bool fast = !__any(run && !ray.fin);   // wave-uniform
#ifdef BVH_WIDE_PROFILE
        prof_steps += WIDE_INNER_STEPS;
        pu[9]++; pu[10] += fast ? 0 : 1;
        pu[8] += WIDE_INNER_STEPS * (unsigned long long)__popcll(__ballot(run));
#endif
        for (int s = 0; s < WIDE_INNER_STEPS; s++) {
#ifdef BVH_WIDE_PROFILE
            {
                const bool in = (cur & WIDE_INNER) != 0u;
                pu[0]++; pu[1] += __popcll(__ballot(in));
                pu[2] += __any(in && (cur & WIDE_RESIDENT)) ? 1 : 0;
                pu[3] += __any(in && !(cur & WIDE_RESIDENT)) ? 1 : 0;
                const bool slow = in && sp + 3u > stack_lds;
                pu[4] += __any(slow) ? 1 : 0; pu[5] += __popcll(__ballot(slow));
                const bool lf = !in && cur < CUR_NONE;
                pu[6] += __popcll(__ballot(lf)); pu[7] += __any(lf) ? 1 : 0;
                pu[11] += __any((in || lf) && sp > stack_lds) ? 1 : 0;
            }
#endif
            if (cur & WIDE_INNER) {   // (CUR_NONE and shape indices have bit 31 clear)
                const uint32_t id = cur & (WIDE_RESIDENT - 1u);
                WideRegs<T> nd;
                if (cur & WIDE_RESIDENT) nd = WideIo<T>::from_lds(nodes, id);
                else nd = WideIo<T>::from_global(wide + id);
                const uint32_t m = fast ? wide_hits<T, false>(ray.o, ray.inv, nd) : wide_hits<T, true>(ray.o, ray.inv, nd);
                // the lowest hit slot is visited now, the others wait on the stack, highest slot first
                const uint32_t first = m & (0u - m);
                const uint32_t rest = m ^ first;
#ifdef BVH_WIDE_PROFILE
                pl_boxes += (uint32_t)__popc(m); pl_nohit += m == 0u ? 1u : 0u;
#endif
                const uint32_t s0 = (rest & 8u) ? nd.ref[3] : ((rest & 4u) ? nd.ref[2] : nd.ref[1]);
                const uint32_t s1 = ((rest & 12u) == 12u) ? nd.ref[2] : nd.ref[1];
                if (sp + 3u <= stack_lds) {   // room for three: store them all, count what is real
                    uint32_t* at = s_stack + sp * SB + tid;
                    at[0] = s0; at[SB] = s1; at[2 * SB] = nd.ref[1];
                    sp += (uint32_t)__popc(rest);
                } else {
                    if (rest & 8u) push_slow(nd.ref[3]);
                    if (rest & 4u) push_slow(nd.ref[2]);
                    if (rest & 2u) push_slow(nd.ref[1]);
                }
                cur = first == 0u ? pop_or_none() : (first == 1u ? nd.ref[0] : (first == 2u ? nd.ref[1] : (first == 4u ? nd.ref[2] : nd.ref[3])));
            }
            bool rec = cur < CUR_NONE;   // a leaf: report it, take the next pending grandchild
            const uint32_t shape = cur;
            if (rec) cur = pop_or_none();
            if (GUIDE) {   // a leaf candidate of the guide walk: the shape's own f64 box and the f64 ray decide (wave-uniform skip: candidates are rare)
                if (__any(rec)) {
                    if (rec) rec = guide_leaf_hit(ga, ITEMS_LOG4 == 0 ? ray.r : (ray.r >> WIDE_ITEM_BITS), shape);
                }
            }
            if (MODE == MODE_INDICES && ITEMS_LOG4 == 0 && w.raybuf) {   // (wave-uniform) the ray's first hits need no record: see WalkOut::raybuf
                if (rec && (ray.cnt >> w.stage_shift) == 0u) {
                    w.raybuf[((size_t)ray.r << w.stage_shift) | ray.cnt] = shape;
                    ray.cnt++;
                    rec = false;
                }
            }
            if constexpr (GUIDE_CLOSEST) {   // (wave-uniform skip, like the f64 box test above: candidates are rare)
                if (__any(rec)) {
                    if (rec) {
                        const double dist = guide_candidate_distance(ga.rays64, ga.tris64, ITEMS_LOG4 == 0 ? ray.r : (ray.r >> WIDE_ITEM_BITS), shape);
                        if (dist < gbest) { gbest = dist; ray.best_prim = shape; }
                        ray.cnt++;
                    }
                }
            } else if (MODE == MODE_INDICES && ITEMS_LOG4 == 0 && w.pool_pair) report_pair(rec, false, shape, ray, pair_pend, w.pool_pair, w.pool_cap, w.ctr, pc, lane, lt);   // (wave-uniform)
            else report<T, MODE>(rec, shape, (T)0, (T)0, ray, w, pc, lane, lt);
        }
        if (ovf) { cur = CUR_NONE; sp = 0; }
    }
    if (__any(ovf) && lane == 0) atomicOr(overflow, 4u);
    if constexpr (GUIDE != 0) { if (__any(guide_bad) && lane == 0) atomicOr(overflow, (uint32_t)WALK_FLAG_GUIDE_RANGE); }
    if (MODE == MODE_INDICES && ITEMS_LOG4 == 0 && w.pool_pair) {
        pool_invalidate_tail16(w.pool_pair, w.pool_cap, pc, lane);
        pc.left = 0;                                                // (nothing left for the epilogue's HitRec form to invalidate)
    }
    walk_epilogue<T, MODE>(w, pc, lane, false, 0, 0, 0, 0);
    if (MODE != MODE_CLOSEST && w.scan_sums) {   // every wave of the workgroup gets here: all items of its rays have retired
        __syncthreads();
        // one atomic per 64-ray block that has hits, on the sum of its scan block (workgroups finish at different times and a
        // scan block's 16 sums come from 16 workgroups: nothing like the per-item atomics that were tried first)
        for (uint32_t b = tid; b < my_blocks; b += bd)
            if (s_bsum[b]) atomicAdd(&w.scan_sums[(b * gridDim.x + blockIdx.x) / (uint32_t)(SCAN_BLOCK / 64)], s_bsum[b]);
    }
#ifdef BVH_WIDE_PROFILE
    if (lane == 0) {
        const size_t wv = gid >> 6;
        if (wv < 16384) {
            g_wide_prof[4 * wv] = prof_t0; g_wide_prof[4 * wv + 1] = prof_t1; g_wide_prof[4 * wv + 2] = wall_clock64(); g_wide_prof[4 * wv + 3] = prof_steps;
            for (int i = 0; i < 12; i++) g_wide_util[16 * wv + i] = pu[i];
            g_wide_util[16 * wv + 12] = 0; g_wide_util[16 * wv + 13] = 0;
        }
    }
    __syncthreads();
    if ((gid >> 6) < 16384) { atomicAdd(&g_wide_util[16 * (gid >> 6) + 12], (unsigned long long)pl_boxes); atomicAdd(&g_wide_util[16 * (gid >> 6) + 13], (unsigned long long)pl_nohit); }
#endif
}

// ---- exclusive scan of per-ray counts ----------------------------------------------------------

// KIND 1 (pair): every ray was walked as two items (k_traverse_lds split): its count is counts[2r] + counts[2r+1].
// KIND 2 (wide walk): counts[r] is non-zero only for rays with hits and ray_items[r] holds the set of the ray's items that
// reported some; k_scan_final copies that set to ray_mask[r] (for the scatter) and puts the zeros back, so both arrays are
// all zero again for the next batch (the walk then stores nothing for the rays — most of them on a sparse scene — that
// hit nothing).
constexpr int COUNT_PLAIN = 0, COUNT_PAIR = 1, COUNT_MASKED = 2;
template <int KIND> __device__ __forceinline__ uint32_t ray_count(const uint32_t* __restrict__ counts, uint32_t r) {
    if (KIND == COUNT_PLAIN || KIND == COUNT_MASKED) return counts[r];
    const uint2 c = reinterpret_cast<const uint2*>(counts)[r];
    return c.x + c.y;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ counts, uint32_t n,
                                                     unsigned long long* __restrict__ blocksums) {
    __shared__ unsigned long long ws[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += (base + j < n) ? ray_count<KIND>(counts, base + j) : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if (lane_id() == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) blocksums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(1024) void k_scan_sums(unsigned long long* __restrict__ blocksums, uint32_t nb,
                                                    unsigned long long* __restrict__ total_out) {
    // exclusive scan of the block sums by ONE workgroup: every thread adds up a contiguous share serially, one 1024-wide scan over
    // the shares, every thread writes its share's prefixes.  (Round 2 looped a 256-wide Hillis-Steele scan with 16 barriers per 256
    // sums: 44 µs for the 2 442 sums of a 10 M-ray batch, 54 µs at 12.5 M — a tenth of the CSR assembly; this form takes ~5 µs.)
    __shared__ unsigned long long ws[16];
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t lo = min(nb, threadIdx.x * per), hi = min(nb, lo + per);
    unsigned long long s = 0;
    for (uint32_t j = lo; j < hi; j++) s += blocksums[j];
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
    unsigned long long inc = s;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const unsigned long long u = __shfl_up(inc, d);
        if (lane >= d) inc += u;
    }
    if (lane == WAVE - 1) ws[wv] = inc;
    __syncthreads();
    unsigned long long run = inc - s;
    for (int w2 = 0; w2 < wv; w2++) run += ws[w2];
    for (uint32_t j = lo; j < hi; j++) { const unsigned long long v = blocksums[j]; blocksums[j] = run; run += v; }
    if (threadIdx.x == 1023u) *total_out = run;   // (the last thread's running sum ends at the total, whether it owns sums or not)
}

constexpr uint32_t SCAN_FUSED_MAX_BLOCKS = 2048;   // up to this many blocks every block sums its predecessors itself

// counts → offsets.  PREFIXED: blocksums already hold exclusive prefixes (k_scan_sums ran, large batches); otherwise
// they are the raw per-block sums of k_scan_reduce and this block adds up its predecessors (one kernel less).
template <int KIND, bool PREFIXED>
__global__ __launch_bounds__(256) void k_scan_final(const uint32_t* __restrict__ counts, uint32_t n,
                                                    const unsigned long long* __restrict__ blocksums,
                                                    unsigned long long* __restrict__ total,
                                                    uint32_t* __restrict__ offsets, uint32_t* __restrict__ ray_items,
                                                    uint16_t* __restrict__ ray_mask, unsigned long long* __restrict__ host_page,
                                                    unsigned long long* __restrict__ other_ctr, const uint32_t* __restrict__ scan_sums,
                                                    uint32_t* __restrict__ other_bsum, uint32_t bsum_cap) {
    __shared__ uint32_t ws[4];
    __shared__ unsigned long long wb[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    const int lane = lane_id();
    unsigned long long before = 0;
    if (PREFIXED) {
        before = blocksums[blockIdx.x];
    } else {
        unsigned long long part = 0;
        if (scan_sums) {   // the walk left the sums (u32) in this batch's set; the other set is zeroed here for the next batch
            for (uint32_t j = threadIdx.x; j < blockIdx.x; j += 256) part += scan_sums[j];
            if (threadIdx.x == 0) other_bsum[blockIdx.x] = 0u;
            if (blockIdx.x == gridDim.x - 1)   // (a previous, larger batch may have left more behind)
                for (uint32_t j = gridDim.x + threadIdx.x; j < bsum_cap; j += 256) other_bsum[j] = 0u;
        } else {
            for (uint32_t j = threadIdx.x; j < blockIdx.x; j += 256) part += blocksums[j];
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d);
        if (lane == 0) wb[threadIdx.x >> 6] = part;
    }
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) { v[j] = (base + j < n) ? ray_count<KIND>(counts, base + j) : 0u; s += v[j]; }
    if (KIND == COUNT_MASKED) {   // rays with hits: keep the item mask for the scatter, zero the word for the next batch
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {
            if (v[j]) {
                if (ray_items) { ray_mask[base + j] = (uint16_t)ray_items[base + j]; ray_items[base + j] = 0u; }
                const_cast<uint32_t*>(counts)[base + j] = 0u;
            }
        }
    }
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
        // With the total every counter of the batch is final (the scatter only reads them): they go to the result's pinned
        // host page from here, and the OTHER counter set — the previous batch's, whose scatter is long done — is zeroed for
        // the next batch.  k_publish_counters as a launch of its own cost 3.8 µs per batch.  (total = ctr[3] of this set.)
        if (host_page) {
            const unsigned long long* ctr = total - 3;
#pragma unroll
            for (int k = 0; k < 8; k++) { host_page[k] = k == 3 ? t : ctr[k]; other_ctr[k] = 0ull; }
            __threadfence_system();
        }
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

// wide walk: a record's `ray` is an item (ray << 5 | j, or the ray itself with one item per ray); the records of item j follow
// those of the ray's earlier items that reported hits (ray_mask) — item_cnt is only valid for those
template <typename T, int NV, int ITEMS_LOG4>
__global__ __launch_bounds__(256) void k_hits_scatter_wide(const HitRec* __restrict__ pool, const T* __restrict__ pool_v,
                                                           const unsigned long long* __restrict__ ctr,
                                                           unsigned long long pool_cap, unsigned long long idx_cap, const uint32_t* __restrict__ offsets,
                                                           const uint32_t* __restrict__ item_cnt, const uint16_t* __restrict__ ray_mask,
                                                           uint32_t* __restrict__ indices, T* __restrict__ vals) {
    const unsigned long long n = ctr[0];
    if (n > pool_cap || ctr[3] > idx_cap) return;  // pool overflowed / more hits than indices[] holds: the host grows it and replays
    for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < n;
         j += (unsigned long long)gridDim.x * blockDim.x) {
        const HitRec h = pool[j];
        if (h.ray == NONE) continue;   // unused tail of a per-wave chunk
        uint32_t d;
        if (ITEMS_LOG4 == 0) {
            d = offsets[h.ray] + h.k;
        } else {
            const uint32_t ray = h.ray >> WIDE_ITEM_BITS, it = h.ray & ((1u << WIDE_ITEM_BITS) - 1u);
            d = offsets[ray] + h.k;
            if (it) {
                const uint32_t mask = ray_mask[ray];
                for (uint32_t i = 0; i < it; i++)
                    if (mask & (1u << i)) d += item_cnt[((size_t)ray << (2 * ITEMS_LOG4)) + i];
            }
        }
        indices[d] = h.shape;
#pragma unroll
        for (int k = 0; k < NV; k++) vals[NV * (size_t)d + k] = pool_v[NV * j + k];
    }
}

// the pair records of a whole-ray index batch (WalkOut::pool_pair) → indices[offsets[ray] + k], + k + 1
__device__ __forceinline__ void scatter_pair_role(uint32_t block, uint32_t nblocks, const uint4* __restrict__ pool, const unsigned long long* __restrict__ ctr,
                                              unsigned long long pool_cap, unsigned long long idx_cap, const uint32_t* __restrict__ offsets,
                                              uint32_t* __restrict__ indices) {
    const unsigned long long n = ctr[0];
    if (n > pool_cap || ctr[3] > idx_cap) return;   // too small: the host grows and replays
    // SCATTER8_UNROLL records per thread and round, loads first: a record costs two dependent reads (the record, then its ray's offset)
    // and the kernel is latency-bound (PMC: 0.40 of the HBM rate, 82 % of wave-time waiting with one record in flight per thread)
    constexpr uint32_t U = SCATTER8_UNROLL;
    const unsigned long long span = (unsigned long long)blockDim.x * U;
    for (unsigned long long j0 = block * span + threadIdx.x; j0 < n; j0 += (unsigned long long)nblocks * span) {
        uint4 h[U];
        uint32_t o[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const unsigned long long j = j0 + (unsigned long long)u * blockDim.x;
            h[u] = j < n ? pool[j] : make_uint4(NONE, 0u, 0u, 0u);   // (NONE also marks the unused tail of a per-wave chunk)
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) o[u] = h[u].x != NONE ? offsets[h[u].x] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < U; u++)
            if (h[u].x != NONE) {
                // both hits in ONE 8-byte store (dword alignment is all a global dwordx2 store needs).  The kernel is bound by its scattered
                // stores, not by the record or offset loads (configs[3] shard: 267 µs; loads alone 115; stores to computed addresses, no
                // offset gather, 270): −2 … −3 % of the assembly
                if (h[u].w != NONE) *reinterpret_cast<DwordPair*>(indices + o[u] + h[u].y) = DwordPair{h[u].z, h[u].w};
                else indices[o[u] + h[u].y] = h[u].z;
            }
    }
}
__global__ __launch_bounds__(256) void k_hits_scatter_pair(const uint4* __restrict__ pool, const unsigned long long* __restrict__ ctr,
                                                       unsigned long long pool_cap, unsigned long long idx_cap, const uint32_t* __restrict__ offsets,
                                                       uint32_t* __restrict__ indices) {
    scatter_pair_role(blockIdx.x, gridDim.x, pool, ctr, pool_cap, idx_cap, offsets, indices);
}

// Staged hits (WalkOut::raybuf) → CSR: one thread per ray copies the ray's first min(count, 2^shift) shapes from its own 2^shift-word
// slot to indices[offsets[ray] ..]: reads of whole 16-byte quads of the slot, writes that neighbouring threads make contiguous.  The later
// hits of a ray (k >= 2^shift) are pool records — pair records through k_hits_scatter_pair / k_hits_scatter8 (BVHGPU_TUNE_WIDE_REC8, the
// default), else 12-byte records through k_hits_scatter_wide.  Together they replace the
// 12-byte-record round trip (write, read, scatter) that cost configs[2] 0.42 ms for 58.8 M hits.  (Fusing this copy into
// k_scan_final — the thread that computes a ray's offset copies its shapes — was measured and dropped: four rays per thread
// break the contiguity of the writes, 0.31 ms against 0.13 + 0.03.)
#ifndef GATHER_RAYS
#define GATHER_RAYS 1
#endif
#ifndef BVH_GATHER_LDS
#define BVH_GATHER_LDS 1
#endif
template <int SHIFT>
__global__ __launch_bounds__(256) void k_hits_gather_staged(const uint32_t* __restrict__ raybuf, const uint32_t* __restrict__ offsets, uint32_t n_rays,
                                                            const unsigned long long* __restrict__ ctr, unsigned long long idx_cap,
                                                            uint32_t* __restrict__ indices) {
    constexpr uint32_t CAP = 1u << SHIFT, R = GATHER_RAYS;   // R rays per thread, their loads issued together (the copy is latency-bound: 0.49 of the HBM rate)
    if (ctr[3] > idx_cap) return;   // more hits than indices[] holds: the host grows it and replays
    uint32_t o0[R], cnt[R];
#pragma unroll
    for (uint32_t u = 0; u < R; u++) {
        const uint32_t r = (blockIdx.x * R + u) * blockDim.x + threadIdx.x;
        o0[u] = 0u; cnt[u] = 0u;
        if (r < n_rays) { o0[u] = offsets[r]; cnt[u] = offsets[r + 1] - o0[u]; }
    }
    uint32_t v[R][CAP];
#pragma unroll
    for (uint32_t u = 0; u < R; u++) {
        const uint32_t r = (blockIdx.x * R + u) * blockDim.x + threadIdx.x;
        const uint4* src = reinterpret_cast<const uint4*>(raybuf + ((size_t)r << SHIFT));
#pragma unroll
        for (uint32_t q = 0; q < CAP / 4; q++) {
            if (4u * q < cnt[u]) { const uint4 x = src[q]; v[u][4 * q] = x.x; v[u][4 * q + 1] = x.y; v[u][4 * q + 2] = x.z; v[u][4 * q + 3] = x.w; }
        }
    }
#if BVH_GATHER_LDS
    // The workgroup's rays are consecutive, so their CSR ranges form ONE contiguous span of indices[] (≈ 6 shapes x 256 rays = 6 KB on
    // configs[2]).  Written straight from the lanes, a store instruction scatters 64 dwords over that span and the L2 evicts partial
    // lines (PMC: 353 MB written for 165 MB of hits); staged through LDS the span goes out as whole 256-byte rows.  Positions k >= CAP
    // of a long ray are not the slot's: they are left out here and written by k_hits_scatter_pair (which runs behind this kernel).
    if (R == 1) {
        constexpr uint32_t SPAN_MAX = 4096;                     // entries of the staging buffer (16 KB); a denser workgroup stores directly
        __shared__ uint32_t s_out[SPAN_MAX];
        __shared__ uint32_t s_base, s_span;
        const uint32_t r0 = blockIdx.x * blockDim.x;
        if (threadIdx.x == 0) {
            const uint32_t r1 = min(r0 + blockDim.x, n_rays);
            s_base = r0 < n_rays ? offsets[r0] : 0u;
            s_span = r0 < n_rays ? offsets[r1] - s_base : 0u;
        }
        __syncthreads();
        const uint32_t base = s_base, span = s_span;
        if (span <= SPAN_MAX) {                                 // (workgroup-uniform)
            for (uint32_t p = threadIdx.x; p < span; p += blockDim.x) s_out[p] = NONE;
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < CAP; k++)
                if (k < cnt[0]) s_out[o0[0] - base + k] = v[0][k];
            __syncthreads();
            for (uint32_t p = threadIdx.x; p < span; p += blockDim.x) {
                const uint32_t x = s_out[p];
                if (x != NONE) indices[base + p] = x;           // (NONE: a long ray's later hits — k_hits_scatter_pair's)
            }
            return;
        }
    }
#endif
#pragma unroll
    for (uint32_t u = 0; u < R; u++) {
#pragma unroll
        for (uint32_t k = 0; k < CAP; k++)
            if (k < cnt[u]) indices[o0[u] + k] = v[u][k];
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

// Name of the walk kernel a batch was handed to, spelled as rocprofv3 prints it (bvhgpu_hits_walk_kernel: bench.py looks its counters
// up under this name instead of rebuilding template strings by hand).  Set by the launch helpers, copied into the result object.
static thread_local char g_walk_kernel[128] = "";
template <typename T> static const char* type_name() { return sizeof(T) == 4 ? "float" : "double"; }

// ------------------------------------------------------------------------------------------------
template <typename T, int MODE, bool STATS>
static void launch_walk(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, const WalkOut<T>& w, bool use_lds,
                        uint32_t split_at) {
    bvhgpu_ctx* ctx = t->ctx;
    hipStream_t st = ctx->stream;
    ensure_flat_arrays(t);   // (a lazy flatten wrote the wide walk's arrays only: the binary array and its LDS slot table follow now)
    const uint32_t n_trav = (uint32_t)t->n_trav;
    const TravNode<T>* nodes = t->trav.as<TravNode<T>>();
    std::snprintf(g_walk_kernel, sizeof g_walk_kernel, "bvhgpu::%s<%s, %d, %s>", use_lds ? "k_traverse_lds" : "k_traverse", type_name<T>(), MODE,
                  STATS ? "true" : "false");
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

// Closest hit of rays that were walked as items (WalkOut::closest_key): the winner's shape comes out of the key, its Intersection is computed
// again from the ray and the triangle — the same function on the same operands as in the walk, hence the same bits (ray_impl.rs:154-213) —
// and the key goes back to all-ones for the next batch.  Rays no item of which met a triangle get {+inf, 0, 0} / NONE (testbase.rs:826-836:
// nothing intersected).
template <typename T>
__global__ __launch_bounds__(256) void k_closest_resolve(unsigned long long* __restrict__ key, const typename Traits<T>::Ray* __restrict__ rays,
                                                         const T* __restrict__ tris, uint32_t n_rays, T* __restrict__ closest, uint32_t* __restrict__ closest_prim) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const unsigned long long k = key[r];
    T out[3] = {Traits<T>::inf(), 0, 0};
    uint32_t prim = NONE;
    if (k != ~0ull) {
        prim = (uint32_t)(k & 0x0FFFFFFFull);
        const typename Traits<T>::Ray* rp = rays + r;
        const T o[3] = {rp->o[0], rp->o[1], rp->o[2]}, d[3] = {rp->d[0], rp->d[1], rp->d[2]};
        ray_triangle<T>(o, d, tris + 9 * (size_t)prim, out);
        key[r] = ~0ull;
    }
    closest[3 * (size_t)r] = out[0]; closest[3 * (size_t)r + 1] = out[1]; closest[3 * (size_t)r + 2] = out[2];
    closest_prim[r] = prim;
}

// closest hit of a ray whose items filed their candidates under (ray, item) (f64): the minimum over the ray's items in item order — items are
// the tree-level-4 subtrees in pre-order, so a later item replaces an earlier one only on a strictly smaller distance, as the reference's
// loop does over its candidate list (testbase.rs:831-833 behind flat_bvh.rs:408).  Leaves the item set all-zero for the next batch.
template <typename T>
__global__ __launch_bounds__(256) void k_closest_resolve_slots(uint32_t* __restrict__ ray_items, const uint32_t* __restrict__ item_prim,
                                                               const typename Traits<T>::Ray* __restrict__ rays, const T* __restrict__ tris, uint32_t n_rays,
                                                               T* __restrict__ closest, uint32_t* __restrict__ closest_prim) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    uint32_t m = ray_items[r];
    T best[3] = {Traits<T>::inf(), 0, 0};
    uint32_t prim = NONE;
    if (m) {
        ray_items[r] = 0u;
        const typename Traits<T>::Ray* rp = rays + r;
        const T o[3] = {rp->o[0], rp->o[1], rp->o[2]}, d[3] = {rp->d[0], rp->d[1], rp->d[2]};
        while (m) {
            const int j = __ffs((int)m) - 1;
            m &= m - 1u;
            const uint32_t p = item_prim[((size_t)r << 4) + (uint32_t)j];
            T out[3];
            ray_triangle<T>(o, d, tris + 9 * (size_t)p, out);
            if (out[0] < best[0]) { best[0] = out[0]; best[1] = out[1]; best[2] = out[2]; prim = p; }
        }
    }
    closest[3 * (size_t)r] = best[0]; closest[3 * (size_t)r + 1] = best[1]; closest[3 * (size_t)r + 2] = best[2];
    closest_prim[r] = prim;
}

// closest-hit batch of the guide walk, whole rays: the walk left the nearest shape per ray; its Intersection, recomputed in f64
template <typename T>
__global__ __launch_bounds__(256) void k_closest_from_prim(const uint32_t* __restrict__ closest_prim, const typename Traits<T>::Ray* __restrict__ rays,
                                                           const T* __restrict__ tris, uint32_t n_rays, T* __restrict__ closest) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const uint32_t p = closest_prim[r];
    T out[3] = {Traits<T>::inf(), 0, 0};
    if (p != NONE) {
        const typename Traits<T>::Ray* rp = rays + r;
        const T o[3] = {rp->o[0], rp->o[1], rp->o[2]}, d[3] = {rp->d[0], rp->d[1], rp->d[2]};
        ray_triangle<T>(o, d, tris + 9 * (size_t)p, out);
    }
    closest[3 * (size_t)r] = out[0]; closest[3 * (size_t)r + 1] = out[1]; closest[3 * (size_t)r + 2] = out[2];
}

// ---- wide walk launch ------------------------------------------------------------------------
// Workgroup geometry: `wg_per_cu` workgroups of `threads` share a CU's 160 KB of LDS; each keeps the per-lane stack
// (stack_lds entries x threads x 4 B) and as many top-of-tree wide nodes as fit in the rest.
template <typename T> struct WideGeom {
    uint32_t threads, wg_per_cu, stack_lds, K;
    size_t lds_bytes;
    WideGeom(const bvhgpu_ctx* ctx, bool whole_rays, bool coherent = false) {
        const bool f64 = sizeof(T) == 8;
        const int want_threads = ctx->tune[BVHGPU_TUNE_WIDE_THREADS] > 0 ? ctx->tune[BVHGPU_TUNE_WIDE_THREADS] : (f64 ? 512 : 1024);
        threads = (uint32_t)std::min(f64 ? 512 : 1024, std::max(64, want_threads & ~63));
        wg_per_cu = (uint32_t)std::max(1, std::min(ctx->tune[BVHGPU_TUNE_WIDE_WG_PER_CU] > 0 ? ctx->tune[BVHGPU_TUNE_WIDE_WG_PER_CU] : 2,
                                                   (int)(2048 / threads)));
        // 16 items per ray (short walks below tree level 4): 4 / 6 / 8 / 10 / 12 entries measured, 6; whole rays on the stand-in scene (a lane on the
        // slow push path in 68-93 % of the steps with 6): 4 / 6 / 8 / 10 / 12 / 16 → 1.71 / 1.58 / 1.50 / 1.48 / 1.47 / 1.50 ms for 10 M primary rays,
        // 2.79 / 2.46 / 2.28 / 2.31 / 2.39 / 2.60 ms for a 12.5 M-ray incoherent shard: 8, and 10 for batches the caller calls COHERENT (with 12 steps
        // between refills: 8 / 10 / 12 entries → 1.39 / 1.36 / 1.37 ms)
        stack_lds = (uint32_t)std::max(0, std::min(ctx->tune[BVHGPU_TUNE_WIDE_STACK_LDS] >= 0 ? ctx->tune[BVHGPU_TUNE_WIDE_STACK_LDS] : (whole_rays ? (coherent ? 10 : 8) : 6), 32));
        // static LDS of the kernel: item table (448 / 832 bytes) + block sums (512 bytes)
        const size_t budget = (size_t)(160 * 1024) / wg_per_cu - (f64 ? 1536 : 1024);
        const size_t stack_stride = f64 ? 512 : 1024;   // (the kernel's MAX_THREADS: its stack planes have a fixed stride)
        const size_t fixed = 80 + (size_t)stack_lds * stack_stride * 4;
        const size_t per_slot = (size_t)WideIo<T>::CHUNKS * 16;
        size_t k = budget > fixed + per_slot ? (budget - fixed) / per_slot : 1;
        if (ctx->tune[BVHGPU_TUNE_WIDE_SLOTS] > 0) k = std::min<size_t>(k, (size_t)ctx->tune[BVHGPU_TUNE_WIDE_SLOTS]);
        K = (uint32_t)std::max<size_t>(1, std::min<size_t>(k, WIDE_SLOTS));
        lds_bytes = 80 + (size_t)K * per_slot + (size_t)stack_lds * stack_stride * 4;
    }
};
// Workgroups of the wide walk for a batch: one ray per lane — unless the rays are cut into items (16 per ray: a workgroup's lanes stay busy
// with a quarter of the rays) and the batch is too small to fill the chip's workgroup slots that way: then the rays are spread over all
// slots, down to BVHGPU_TUNE_WIDE_MIN_RAYS_PER_WG rays per workgroup.  (One ray per lane, a small batch takes the time of ONE workgroup's
// 1024 rays whatever its size; bvhgpu_traverse_host_* walks its batches in such chunks.  profiles/r6_walk_size_sweep.log)
inline size_t wide_grid(const bvhgpu_ctx* ctx, uint32_t threads, uint32_t wg_per_cu, size_t n_rays, int items_log4) {
    const size_t slots = (size_t)ctx->n_cu * wg_per_cu;
    size_t per_wg = threads;
    const int min_rays = ctx->tune[BVHGPU_TUNE_WIDE_MIN_RAYS_PER_WG];
    if (items_log4 == 2 && min_rays > 0 && n_rays * 4 <= slots * threads) {   // (measured: 33 K / 66 K / 125 K rays 48 -> 37 / 38 / 41 µs; 250 K rays 49 -> 53)
        const size_t spread = ((n_rays + slots - 1) / slots + 63) & ~(size_t)63;
        per_wg = std::min<size_t>(threads, std::max<size_t>(spread, (size_t)std::max(64, min_rays & ~63)));
    }
    const size_t full = (n_rays + per_wg - 1) / per_wg;
    return std::min<size_t>(std::max<size_t>(full, 1), slots);
}
constexpr uint32_t WIDE_GSTACK = 24;   // stack entries per lane beyond the LDS part, in HBM (a walk pushes at most 3 per wide level)

// GUIDE: T = float on an f64 tree — the nodes are the tree's guide boxes, rays_dev unused (NULL), ga the f64 batch: every ray is converted where the walk loads it (guide_ray_load)
template <typename T, int MODE, int ITEMS_LOG4, int GUIDE = 0>
static void launch_wide(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, const WalkOut<T>& w, bvhgpu_hits* h,
                        uint32_t* ovf_flag, bool early_items, GuideArgs ga = GuideArgs{nullptr, nullptr, nullptr, nullptr}) {
    bvhgpu_ctx* ctx = t->ctx;
    hipStream_t st = ctx->stream;
    const WideGeom<T> g(ctx, ITEMS_LOG4 == 0, (h->flags & BVHGPU_TRAVERSE_COHERENT) != 0);
    const dim3 grid((unsigned)wide_grid(ctx, g.threads, g.wg_per_cu, n_rays, ITEMS_LOG4));
    uint32_t* list = nullptr;
    if (ITEMS_LOG4 > 0) {   // every workgroup's region of the live-item list: its rays x 4^L entries
        const size_t n_blocks = (n_rays + 63) / 64;
        const size_t per_wg = ((n_blocks + grid.x - 1) / grid.x) * 64;   // k_traverse_wide: capacity of a workgroup's share
        h->witems.reserve(((size_t)grid.x * per_wg << (2 * ITEMS_LOG4)) * 4 + 16);
        list = h->witems.as<uint32_t>();
    }
    const uint32_t* wg_items = nullptr;
    if (ITEMS_LOG4 == 2 && early_items) {
        // The item filter runs on the ctx's side stream as soon as the build on the main stream has split tree level 3 (t->ev_top) —
        // beside the remaining level passes and the workgroup / wave tiers, which leave most of the chip idle — and the walk waits
        // for it (h->ev_items) instead of filtering in its own prologue.  k_wide_items checks on the device that the top of the tree
        // is what this needs; if not, every workgroup of the walk filters its rays itself.
        if (!ctx->side) BVH_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        if (!h->ev_items) BVH_HIP(hipEventCreateWithFlags(&h->ev_items, hipEventDisableTiming));
        h->wg_items.reserve((size_t)grid.x * 2 * 4 + 16);
        BVH_HIP(hipStreamWaitEvent(ctx->side, t->ev_top, 0));
        hipLaunchKernelGGL(k_wide_items<T>, grid, dim3(256), 0, ctx->side, t->nodes.as<typename Traits<T>::Node>(), (uint32_t)t->n_nodes,
                           t->node_count.as<uint32_t>(), t->ctr.as<uint32_t>(), rays_dev, (uint32_t)n_rays, list, h->wg_items.as<uint32_t>());
        BVH_HIP(hipEventRecord(h->ev_items, ctx->side));
        BVH_HIP(hipStreamWaitEvent(st, h->ev_items, 0));
        wg_items = h->wg_items.as<uint32_t>();
    }
    const size_t lanes = (size_t)grid.x * g.threads;
    h->wstack.reserve(lanes * WIDE_GSTACK * 4);
    constexpr int MAXT = sizeof(T) == 8 ? 512 : 1024;
    constexpr int MINW = sizeof(T) == 8 ? BVH_WIDE_MIN_WAVES_F64 : BVH_WIDE_MIN_WAVES_F32;
    auto kern = &k_traverse_wide<T, MODE, ITEMS_LOG4, MAXT, MINW, GUIDE>;
    std::snprintf(g_walk_kernel, sizeof g_walk_kernel, "bvhgpu::k_traverse_wide<%s, %d, %d, %d, %d, %d>", type_name<T>(), MODE, ITEMS_LOG4, MAXT, MINW, GUIDE);
    static thread_local size_t lds_attr[16] = {};   // per device: dynamic-LDS limit already set for this instantiation
    size_t& have = lds_attr[ctx->device & 15];
    if (have < g.lds_bytes) {
        BVH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes));
        have = g.lds_bytes;
    }
    hipLaunchKernelGGL(kern, grid, dim3(g.threads), g.lds_bytes, st, GUIDE ? t->wide_guide.as<WideNode<T>>() : t->wide.as<WideNode<T>>(),
                       t->wslot_node.as<uint32_t>(), g.K, g.stack_lds, rays_dev, (uint32_t)n_rays, list, wg_items, w, h->wstack.as<uint32_t>(),
                       WIDE_GSTACK, ovf_flag, ga,
                       (uint32_t)((h->flags & BVHGPU_TRAVERSE_COHERENT) ? BVH_WIDE_INNER_STEPS_COHERENT : BVH_WIDE_INNER_STEPS_WHOLE));
}

// ---- one batch = enqueue (no host round trip) + check (after the stream has been synchronised) --------------------
// What the enqueue decided is kept in the result object, so that the check — and an asynchronous caller's
// bvhgpu_hits_wait — can replay the batch when the hit pool, a lane's heap or a lane's stack was too small.
template <typename T>
void traverse_enqueue(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, unsigned flags, bvhgpu_hits* h) {
    bvhgpu_ctx* ctx = t->ctx;
    hipStream_t st = ctx->stream;
    const bool stats = (flags & BVHGPU_TRAVERSE_STATS) != 0;
    const bool coherent = (flags & BVHGPU_TRAVERSE_COHERENT) != 0;
    const int ordered = (flags & BVHGPU_TRAVERSE_NEAREST_FIRST) ? 1 : ((flags & BVHGPU_TRAVERSE_FARTHEST_FIRST) ? 2 : 0);
    const int mode = (flags & BVHGPU_TRAVERSE_CLOSEST) ? MODE_CLOSEST
                   : (flags & BVHGPU_TRAVERSE_TRIANGLES) ? MODE_TRIANGLES
                   : (flags & BVHGPU_TRAVERSE_T_SLICE) ? MODE_T_SLICE : MODE_INDICES;
    const int nv = mode == MODE_T_SLICE ? 2 : (mode == MODE_TRIANGLES ? 3 : 0);
    const int variant = ctx->tune[BVHGPU_TUNE_TRAVERSE_VARIANT];
    // (the COHERENT hint does not choose the walk kernel — on 10 M primary rays the wide walk takes 1.5 ms, the persistent binary walk 2.2 ms
    //  and one ray per lane per launch 5.4 ms — it chooses how the wide walk hands over its hits: see `staged` below)
    const bool big_batch = !ordered && n_rays >= (size_t)ctx->tune[BVHGPU_TUNE_TRAVERSE_LDS_MIN_RAYS];
    // walk kernel: wide walk (see k_traverse_wide for what it needs), else persistent workgroups over the binary array with
    // its top in LDS, else one ray per lane per launch
    const bool use_wide = variant >= 3 && big_batch && t->has_wide && !t->exact_only && !t->unfolded && !stats &&
                          mode != MODE_T_SLICE && !h->force_binary && t->n >= 2;
    const bool use_lds = !use_wide && variant != 0 && t->slot_entry.p != nullptr && big_batch;
    // several items per ray while a resident lane gets fewer than ~4 rays: the tail of the launch dominates there
    const bool few_rays = n_rays < (size_t)ctx->n_cu * 2048 * 4;
    uint32_t split_at = 0;
    if (use_lds && mode != MODE_CLOSEST && ctx->tune[BVHGPU_TUNE_TRAVERSE_SPLIT] != 0 && t->n >= 2 && !t->unfolded && few_rays)
        split_at = 1;   // the kernel reads the boundary itself: exit index of entry 0 (the root's left child)
    int items_log4 = 0;
    if (use_wide && mode != MODE_CLOSEST && n_rays < WIDE_ITEM_MAX_RAYS) {
        const int want = ctx->tune[BVHGPU_TUNE_WIDE_ITEMS_LOG4];
        items_log4 = want >= 0 ? std::min(want, 2) : (few_rays ? 2 : 0);
    }
    // closest hit: the same cut into 16 items below ~2 M rays — the per-ray minimum over the items goes through WalkOut::closest_key (f32: one
    // 64-bit atomicMin per item with a candidate) or through the (ray, item) slots (f64: k_closest_resolve_slots)
    if (use_wide && mode == MODE_CLOSEST && n_rays < WIDE_ITEM_MAX_RAYS) {
        const int want = ctx->tune[BVHGPU_TUNE_WIDE_ITEMS_LOG4];
        items_log4 = (want >= 0 ? want >= 2 : few_rays) ? 2 : 0;
    }
    // the item filter beside the build (launch_wide): the tree is being rebuilt on this stream, the build has recorded the event behind
    // the pass that splits level 3, and the caller says that the rays do not depend on anything enqueued since
    const bool early_items = use_wide && items_log4 == 2 && (flags & BVHGPU_TRAVERSE_RAYS_READY) != 0 && t->pending_build && t->ev_top != nullptr &&
                             t->ev_top_gen == t->gen && ctx->tune[BVHGPU_TUNE_WIDE_EARLY_ITEMS] != 0;
    // f64 index batches: the f32 walk over the tree's guide boxes, leaf candidates confirmed in f64 (guide_ray_load; a result object that met
    // a ray outside the guide walk's range stays with the f64 walk)
    // A result object that met a ray outside the guide walk's range backs off: the next `guide_skip` f64 index batches take the f64 walk
    // straight away (1, 2, 4 … 64 batches on consecutive failures — a workload of axis-parallel rays pays one wasted guide walk in 65),
    // then the guide is tried again; a batch that stays in range resets the back-off (the fall-back used to be for ever).
    const bool guide_ok = sizeof(T) == 8 && use_wide && (mode == MODE_INDICES || mode == MODE_CLOSEST) && t->has_guide && !early_items && ctx->tune[BVHGPU_TUNE_WIDE_F64_GUIDE] != 0;
    const bool replaying_out_of_range = h->no_guide;      // traverse_check sent this very batch back
    // (one back-off slot per BATCH: a replay of the same batch — pool / index growth, a stack overflow's switch to the binary walk — takes none)
    const bool first_enqueue = h->pend_attempts == 0 && !h->force_binary && !replaying_out_of_range;
    if (guide_ok && first_enqueue && h->guide_skip > 0) h->guide_skip--;
    const bool use_guide = guide_ok && !replaying_out_of_range && h->guide_skip == 0;
    if (replaying_out_of_range) {
        h->no_guide = false;
        h->guide_backoff = h->guide_backoff ? std::min(2 * h->guide_backoff, 64u) : 1u;
        h->guide_skip = h->guide_backoff + 1u;             // (+1: the decrement of the next batch's enqueue)
    }
    h->pend_guide = use_guide;
    const size_t n_items = split_at ? 2 * n_rays : n_rays;
    h->ctx = ctx; h->dtype = Traits<T>::dtype; h->n_rays = n_rays; h->flags = flags; h->total = 0;
    h->stats = bvhgpu_traverse_stats{0, 0, 0, 0, 0};
    h->pend_tree = t; h->pend_rays = rays_dev; h->pend_wide = use_wide; h->pend_unfolded = t->unfolded || t->n == 1;
    // two counter sets: a batch uses one and (k_scan_final) zeroes the other for the batch after it
    if (h->ctr.reserve(16 * sizeof(unsigned long long))) h->ctr_clean = false;
    if (!h->pin) BVH_HIP(hipHostMalloc(&h->pin, 64, hipHostMallocDefault));
    unsigned long long* pin = reinterpret_cast<unsigned long long*>(h->pin);
    unsigned long long* ctr = h->ctr.as<unsigned long long>() + 8 * (h->ctr_set & 1);
    unsigned long long* ctr_other = h->ctr.as<unsigned long long>() + 8 * ((h->ctr_set & 1) ^ 1);

    WalkOut<T> w;
    w.counts = nullptr; w.pool = nullptr; w.pool_v = nullptr; w.pool_cap = 0; w.ctr = ctr;
    w.tris = t->tris.as<T>(); w.closest = nullptr; w.closest_prim = nullptr; w.closest_key = nullptr; w.item_cnt = nullptr; w.ray_items = nullptr; w.scan_sums = nullptr;
    w.raybuf = nullptr; w.stage_shift = 0; w.pool_pair = nullptr;

    uint32_t* ovf_flag = reinterpret_cast<uint32_t*>(ctr + 7);   // bit 0 ordered-iterator stack, bit 1 heap workspace, bit 2 wide-walk stack
    const bool best_first = ordered && (flags & BVHGPU_TRAVERSE_BEST_FIRST) != 0;
    const unsigned heap_grid = (unsigned)std::min<size_t>((n_rays + 255) / 256, (size_t)ctx->n_cu * 4);
    auto launch_ordered = [&](auto mode_tag, auto asc_tag) {
        constexpr int M = decltype(mode_tag)::value;
        constexpr bool A = decltype(asc_tag)::value;
        std::snprintf(g_walk_kernel, sizeof g_walk_kernel, "bvhgpu::%s<%s, %d, %s>", best_first ? "k_traverse_heap" : "k_traverse_ordered", type_name<T>(), M,
                      A ? "true" : "false");
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
    auto dispatch_wide = [&](auto mode_tag) {
        constexpr int M = decltype(mode_tag)::value;
        if constexpr (sizeof(T) == 8 && M == MODE_INDICES) {
            if (use_guide) {
                const bvhgpu_ray_f32* r32 = nullptr;   // (the guide walk converts every f64 ray where it loads it: no f32 copy of the batch)
                WalkOut<float> wg;   // the same outputs: an index batch touches none of the T-typed ones
                wg.counts = w.counts; wg.pool = w.pool; wg.pool_v = nullptr; wg.pool_cap = w.pool_cap; wg.ctr = w.ctr; wg.tris = nullptr;
                wg.closest = nullptr; wg.closest_prim = nullptr; wg.closest_key = nullptr; wg.item_cnt = w.item_cnt; wg.ray_items = w.ray_items; wg.scan_sums = w.scan_sums;
                wg.pool_pair = w.pool_pair; wg.raybuf = w.raybuf; wg.stage_shift = w.stage_shift;
                const GuideArgs ga{reinterpret_cast<const bvhgpu_ray_f64*>(rays_dev), t->aabbs.as<double>(), t->guide_info.as<float>(), nullptr};
                if (items_log4 == 2) launch_wide<float, MODE_INDICES, 2, 1>(t, r32, n_rays, wg, h, ovf_flag, false, ga);
                else if (items_log4 == 1) launch_wide<float, MODE_INDICES, 1, 1>(t, r32, n_rays, wg, h, ovf_flag, false, ga);
                else launch_wide<float, MODE_INDICES, 0, 1>(t, r32, n_rays, wg, h, ovf_flag, false, ga);
                return;
            }
        }
        if constexpr (sizeof(T) == 8 && M == MODE_CLOSEST) {
            if (use_guide) {   // closest hit of an f64 batch: the f32 walk over the guide boxes; every leaf candidate's box AND triangle decided in f64
                WalkOut<float> wg;
                wg.counts = nullptr; wg.pool = nullptr; wg.pool_v = nullptr; wg.pool_cap = 0; wg.ctr = w.ctr; wg.tris = nullptr;
                wg.closest = nullptr; wg.closest_prim = w.closest_prim; wg.closest_key = nullptr; wg.item_cnt = w.item_cnt; wg.ray_items = w.ray_items; wg.scan_sums = nullptr;
                wg.pool_pair = nullptr; wg.raybuf = nullptr; wg.stage_shift = 0;
                const GuideArgs ga{reinterpret_cast<const bvhgpu_ray_f64*>(rays_dev), t->aabbs.as<double>(), t->guide_info.as<float>(), t->tris.as<double>()};
                if (items_log4 == 2) launch_wide<float, MODE_CLOSEST, 2, 1>(t, nullptr, n_rays, wg, h, ovf_flag, false, ga);
                else launch_wide<float, MODE_CLOSEST, 0, 1>(t, nullptr, n_rays, wg, h, ovf_flag, false, ga);
                return;
            }
        }
        if (M != MODE_CLOSEST && items_log4 == 2) launch_wide<T, M, (M == MODE_CLOSEST ? 0 : 2)>(t, rays_dev, n_rays, w, h, ovf_flag, early_items);
        else if (M != MODE_CLOSEST && items_log4 == 1) launch_wide<T, M, (M == MODE_CLOSEST ? 0 : 1)>(t, rays_dev, n_rays, w, h, ovf_flag, false);
        else if (M == MODE_CLOSEST && items_log4 == 2) launch_wide<T, M, (M == MODE_CLOSEST ? 2 : 0)>(t, rays_dev, n_rays, w, h, ovf_flag, false);
        else launch_wide<T, M, 0>(t, rays_dev, n_rays, w, h, ovf_flag, false);
    };
#define DISPATCH_WALK_INNER()                                                                        \
    do {                                                                                             \
        if (ordered) {                                                                               \
            switch (mode) {                                                                          \
                case MODE_INDICES: dispatch_ordered(std::integral_constant<int, MODE_INDICES>{}); break;     \
                case MODE_TRIANGLES: dispatch_ordered(std::integral_constant<int, MODE_TRIANGLES>{}); break; \
                default: dispatch_ordered(std::integral_constant<int, MODE_CLOSEST>{}); break;       \
            }                                                                                        \
            break;                                                                                   \
        }                                                                                            \
        if (use_wide) {                                                                              \
            switch (mode) {                                                                          \
                case MODE_INDICES: dispatch_wide(std::integral_constant<int, MODE_INDICES>{}); break;        \
                case MODE_TRIANGLES: dispatch_wide(std::integral_constant<int, MODE_TRIANGLES>{}); break;    \
                default: dispatch_wide(std::integral_constant<int, MODE_CLOSEST>{}); break;          \
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
#define DISPATCH_WALK() do { g_walk_kernel[0] = 0; DISPATCH_WALK_INNER(); h->walk_kernel = g_walk_kernel; } while (0)

    if (!h->ctr_clean) BVH_HIP(hipMemsetAsync(ctr, 0, 8 * sizeof(unsigned long long), st));   // (only this batch's set has to be clean)
    h->ctr_clean = false;
    if (mode == MODE_CLOSEST) {   // no CSR: one Intersection + shape per ray
        h->closest.reserve(std::max<size_t>(n_rays, 1) * 3 * sizeof(T));
        h->closest_prim.reserve(std::max<size_t>(n_rays, 1) * 4);
        if (n_rays == 0) { h->pend_tree = nullptr; return; }
        w.closest = h->closest.as<T>(); w.closest_prim = h->closest_prim.as<uint32_t>();
        const bool by_items = use_wide && items_log4 == 2;
        if (by_items && sizeof(T) == 4) {   // the per-ray keys: all-ones between batches (k_closest_resolve puts them back)
            if (h->closest_key.reserve(n_rays * sizeof(unsigned long long))) h->ckey_clean = false;
            if (!h->ckey_clean) BVH_HIP(hipMemsetAsync(h->closest_key.p, 0xFF, h->closest_key.cap, st));
            h->ckey_clean = true;
            w.closest_key = h->closest_key.as<unsigned long long>();
        } else if (by_items) {   // f64: candidates by (ray, item), the rays' item sets all-zero between batches (k_closest_resolve_slots puts the zeros back)
            h->item_cnt.reserve(((n_rays << 4) + 1) * 4);
            if (h->ray_items.reserve((n_rays + 1) * 4)) BVH_HIP(hipMemsetAsync(h->ray_items.p, 0, h->ray_items.cap, st));
            w.item_cnt = h->item_cnt.as<uint32_t>();
            w.ray_items = h->ray_items.as<uint32_t>();
        }
        if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[4], st)); }
        DISPATCH_WALK();
        if (ctx->timing) BVH_HIP(hipEventRecord(ctx->ev[5], st));
        if (by_items && sizeof(T) == 4)
            hipLaunchKernelGGL(k_closest_resolve<T>, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, w.closest_key, rays_dev, t->tris.as<T>(),
                               (uint32_t)n_rays, w.closest, w.closest_prim);
        else if (by_items)
            hipLaunchKernelGGL(k_closest_resolve_slots<T>, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, w.ray_items, (const uint32_t*)w.item_cnt, rays_dev,
                               t->tris.as<T>(), (uint32_t)n_rays, w.closest, w.closest_prim);
        else if (use_guide)   // whole rays of the guide walk: the shapes are in closest_prim, their Intersections follow
            hipLaunchKernelGGL(k_closest_from_prim<T>, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, (const uint32_t*)w.closest_prim, rays_dev, t->tris.as<T>(),
                               (uint32_t)n_rays, w.closest);
        if (ctx->timing) BVH_HIP(hipEventRecord(ctx->ev[6], st));
        hipLaunchKernelGGL(k_publish_counters, dim3(1), dim3(64), 0, st, ctr, pin);   // readback + reset for the next call
        h->ctr_clean = true;
        join_flat(t);
        return;
    }

    h->offsets.reserve((n_rays + 1) * 4);
    const uint32_t nb = (uint32_t)((n_rays + SCAN_BLOCK - 1) / SCAN_BLOCK);
    if (h->blocksums.reserve((nb + 1) * sizeof(unsigned long long))) h->bs_clean = false;
    if (h->pool_cap == 0) {
        h->pool_cap = std::max<size_t>(n_rays, (size_t)1 << 16);
        // pair records take 16 bytes whether they hold one hit or two: a first pool sized in 12-byte slots would hold 0.75 records per ray and
        // a batch of single-hit rays would overflow it where the 12-byte records fitted — size it so that every ray has a record
        if (use_wide && items_log4 == 0 && mode == MODE_INDICES && ctx->tune[BVHGPU_TUNE_WIDE_REC8] != 0) h->pool_cap = (h->pool_cap * 16 + sizeof(HitRec) - 1) / sizeof(HitRec);
    }
    if (n_rays == 0) {
        BVH_HIP(hipMemsetAsync(h->offsets.p, 0, 4, st));
        h->pend_tree = nullptr;
        return;
    }
    // staged output (WalkOut::raybuf): whole rays, indices only — the hit-heavy large batches; stage_shift 3 = eight shapes per ray
    // Default: for batches the caller calls COHERENT (primary rays).  Measured on the stand-in scene: 10 M coherent rays 2.60 → 2.17 ms per
    // step (the walk itself 1.79 → 1.54: no chunk atomics, a third of the bytes; neighbouring rays' slots are written close in time and
    // merge in L2); 12.5 M incoherent rays 3.30 → 3.23; 100 M incoherent rays 21.1 → 22.0 (the walk 16.3 → 18.4: every hit of a
    // wave lands on a line of its own) — so incoherent batches keep the pool, whose records a wave writes contiguously.
    const int stage_knob = ctx->tune[BVHGPU_TUNE_WIDE_STAGE_SHIFT];
    const int stage_shift = stage_knob < 0 ? (coherent ? 3 : 0) : std::min(stage_knob, 5);
    const bool staged = use_wide && items_log4 == 0 && mode == MODE_INDICES && stage_shift >= 2;
    h->pend_staged = staged;
    if (h->idx_cap < h->pool_cap) h->idx_cap = h->pool_cap;
    h->pool.reserve(h->pool_cap * sizeof(HitRec));
    h->indices.reserve(h->idx_cap * 4);
    if (staged) { h->raybuf.reserve(((size_t)n_rays << stage_shift) * 4 + 64); w.raybuf = h->raybuf.as<uint32_t>(); w.stage_shift = (uint32_t)stage_shift; }
    // pair records (8 bytes per hit) for whole-ray index batches
    const bool rec8 = use_wide && items_log4 == 0 && mode == MODE_INDICES && ctx->tune[BVHGPU_TUNE_WIDE_REC8] != 0;
    h->pend_rec8 = rec8;
    if (nv) {
        h->pool_t.reserve(h->pool_cap * nv * sizeof(T));
        (mode == MODE_T_SLICE ? h->tslice : h->isect).reserve(h->pool_cap * nv * sizeof(T));
    }
    // (the pool's bytes then hold 16-byte pair records: three for every four HitRec slots)
    const unsigned long long cap = rec8 ? (unsigned long long)h->pool_cap * sizeof(HitRec) / 16u : h->pool_cap;
    uint32_t* counts;
    uint32_t* bsum_other = nullptr;
    if (use_wide) {   // per-ray words kept all-zero between batches (k_scan_final puts the zeros back)
        if (h->wcounts.reserve((n_rays + 1) * 4)) h->wcounts_clean = false;
        if (!h->wcounts_clean) BVH_HIP(hipMemsetAsync(h->wcounts.p, 0, h->wcounts.cap, st));
        h->wcounts_clean = false;
        if (items_log4) {
            h->ray_mask.reserve((n_rays + 1) * 2);
            h->item_cnt.reserve(((n_rays << (2 * items_log4)) + 1) * 4);
            if (h->ray_items.reserve((n_rays + 1) * 4)) BVH_HIP(hipMemsetAsync(h->ray_items.p, 0, h->ray_items.cap, st));   // then kept zero like wcounts
            w.item_cnt = h->item_cnt.as<uint32_t>();
            w.ray_items = h->ray_items.as<uint32_t>();
        }
        counts = h->wcounts.as<uint32_t>();
        // the walk's workgroups leave the hits per 64-ray block (their own blocks: LDS sums): no reduce pass for the scan
        {
            const WideGeom<T> gt(ctx, items_log4 == 0, coherent);
            const WideGeom<float> gf(ctx, items_log4 == 0, coherent);   // (the guide walk of an f64 batch launches the f32 geometry)
            const uint32_t g_threads = use_guide ? gf.threads : gt.threads, g_wg_per_cu = use_guide ? gf.wg_per_cu : gt.wg_per_cu;
            const size_t grid = wide_grid(ctx, g_threads, g_wg_per_cu, n_rays, items_log4);   // launch_wide: the same
            const size_t n_blocks = (n_rays + 63) / 64;
            if (nb <= SCAN_FUSED_MAX_BLOCKS && (n_blocks + grid - 1) / grid <= WIDE_BSUM_MAX) {
                // two sets of SCAN_FUSED_MAX_BLOCKS sums, used alternately like the counter sets (k_scan_final zeroes the other one)
                if (h->scan_sums.reserve(2 * SCAN_FUSED_MAX_BLOCKS * 4)) BVH_HIP(hipMemsetAsync(h->scan_sums.p, 0, h->scan_sums.cap, st));
                w.scan_sums = h->scan_sums.as<uint32_t>() + (size_t)SCAN_FUSED_MAX_BLOCKS * (h->bsum_set & 1);
                bsum_other = h->scan_sums.as<uint32_t>() + (size_t)SCAN_FUSED_MAX_BLOCKS * ((h->bsum_set & 1) ^ 1);
                h->bsum_set ^= 1;   // (this batch's k_scan_final zeroes the other set: the next batch's)
            }
        }
    } else {
        h->counts.reserve((n_items + 1) * 4);
        counts = h->counts.as<uint32_t>();
    }
    w.counts = counts; w.pool = h->pool.as<HitRec>(); w.pool_v = h->pool_t.as<T>(); w.pool_cap = cap;
    w.pool_pair = rec8 ? h->pool.as<uint4>() : nullptr;
    if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[4], st)); }
    DISPATCH_WALK();
    if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[5], st)); }
    unsigned long long* bs = h->blocksums.as<unsigned long long>();
    uint32_t* offs = h->offsets.as<uint32_t>();
    uint16_t* rmask = h->ray_mask.as<uint16_t>();
    uint32_t* ritems = (use_wide && items_log4) ? h->ray_items.as<uint32_t>() : nullptr;
    const uint32_t nr = (uint32_t)n_rays;
    const int kind = use_wide ? COUNT_MASKED : (split_at ? COUNT_PAIR : COUNT_PLAIN);
    auto scan = [&](auto kind_tag) {
        constexpr int KD = decltype(kind_tag)::value;
        if (!w.scan_sums) hipLaunchKernelGGL(k_scan_reduce<KD>, dim3(nb), dim3(256), 0, st, counts, nr, bs);
        if (nb > SCAN_FUSED_MAX_BLOCKS) {
            hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, bs, nb, ctr + 3);
            hipLaunchKernelGGL((k_scan_final<KD, true>), dim3(nb), dim3(256), 0, st, counts, nr, bs, ctr + 3, offs, ritems, rmask,
                               (unsigned long long*)nullptr, (unsigned long long*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
        } else {
            hipLaunchKernelGGL((k_scan_final<KD, false>), dim3(nb), dim3(256), 0, st, counts, nr, bs, ctr + 3, offs, ritems, rmask, pin, ctr_other,
                               (const uint32_t*)w.scan_sums, bsum_other, (uint32_t)SCAN_FUSED_MAX_BLOCKS);
        }
    };
    if (kind == COUNT_MASKED) scan(std::integral_constant<int, COUNT_MASKED>{});
    else if (kind == COUNT_PAIR) scan(std::integral_constant<int, COUNT_PAIR>{});
    else scan(std::integral_constant<int, COUNT_PLAIN>{});
    if (use_wide) h->wcounts_clean = true;   // (stays true only if the check finds that the batch ran to completion)
    const uint32_t* pair_counts = split_at ? counts : nullptr;
    const int sgrid = (int)std::min<size_t>((cap + 255) / 256, (size_t)ctx->n_cu * 8);
    T* vals = mode == MODE_T_SLICE ? h->tslice.as<T>() : h->isect.as<T>();
    uint32_t* indices = h->indices.as<uint32_t>();
    if (staged) {   // the rays' first 2^shift shapes, straight from their slots; the pool records (later hits) follow below
        const unsigned ggrid = (unsigned)((n_rays + 256 * GATHER_RAYS - 1) / (256 * GATHER_RAYS));
        const unsigned long long icap = h->idx_cap;
        switch (stage_shift) {
            case 2: hipLaunchKernelGGL(k_hits_gather_staged<2>, dim3(ggrid), dim3(256), 0, st, w.raybuf, offs, nr, ctr, icap, indices); break;
            case 3: hipLaunchKernelGGL(k_hits_gather_staged<3>, dim3(ggrid), dim3(256), 0, st, w.raybuf, offs, nr, ctr, icap, indices); break;
            case 4: hipLaunchKernelGGL(k_hits_gather_staged<4>, dim3(ggrid), dim3(256), 0, st, w.raybuf, offs, nr, ctr, icap, indices); break;
            default: hipLaunchKernelGGL(k_hits_gather_staged<5>, dim3(ggrid), dim3(256), 0, st, w.raybuf, offs, nr, ctr, icap, indices); break;
        }
    }
    if (rec8) {
        hipLaunchKernelGGL(k_hits_scatter_pair, dim3(sgrid), dim3(256), 0, st, w.pool_pair, ctr, cap, (unsigned long long)h->idx_cap, offs, indices);
    } else if (use_wide) {
        const uint32_t* icnt = h->item_cnt.as<uint32_t>();
#define SCATTER_WIDE(NV, L4) hipLaunchKernelGGL((k_hits_scatter_wide<T, NV, L4>), dim3(sgrid), dim3(256), 0, st, w.pool, w.pool_v, ctr, cap, (unsigned long long)h->idx_cap, offs, icnt, rmask, indices, vals)
        if (nv == 3) { if (items_log4 == 2) SCATTER_WIDE(3, 2); else if (items_log4 == 1) SCATTER_WIDE(3, 1); else SCATTER_WIDE(3, 0); }
        else { if (items_log4 == 2) SCATTER_WIDE(0, 2); else if (items_log4 == 1) SCATTER_WIDE(0, 1); else SCATTER_WIDE(0, 0); }
#undef SCATTER_WIDE
    } else if (nv == 2) {
        hipLaunchKernelGGL((k_hits_scatter<T, 2>), dim3(sgrid), dim3(256), 0, st, w.pool, w.pool_v, ctr, cap, offs, pair_counts, indices, vals);
    } else if (nv == 3) {
        hipLaunchKernelGGL((k_hits_scatter<T, 3>), dim3(sgrid), dim3(256), 0, st, w.pool, w.pool_v, ctr, cap, offs, pair_counts, indices, vals);
    } else {
        hipLaunchKernelGGL((k_hits_scatter<T, 0>), dim3(sgrid), dim3(256), 0, st, w.pool, w.pool_v, ctr, cap, offs, pair_counts, indices, vals);
    }
    if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[6], st)); }
    if (nb > SCAN_FUSED_MAX_BLOCKS) {   // readback + reset for the next call (smaller batches: k_scan_final's last block did both)
        hipLaunchKernelGGL(k_publish_counters, dim3(1), dim3(64), 0, st, ctr, pin);
    } else {
        h->ctr_set ^= 1;   // the set that was just zeroed
    }
    h->ctr_clean = true;
    join_flat(t);   // (BVHGPU_TUNE_FLATTEN_LAZY = 2: the flatten part that ran beside this walk — the batch's wait covers it)
#undef DISPATCH_WALK
}

// After the stream has been synchronised: true = the batch is complete; false = something was too small and has been
// grown (or the walk switched) — enqueue again.  Throws on the conditions the reference would panic on.
bool traverse_check(bvhgpu_hits* h) {
    if (!h->pend_tree) return true;   // nothing was launched (empty batch)
    bvhgpu_ctx* ctx = h->ctx;
    const unsigned flags = h->flags;
    const bool stats = (flags & BVHGPU_TRAVERSE_STATS) != 0;
    const bool ordered = (flags & (BVHGPU_TRAVERSE_NEAREST_FIRST | BVHGPU_TRAVERSE_FARTHEST_FIRST)) != 0;
    const bool best_first = ordered && (flags & BVHGPU_TRAVERSE_BEST_FIRST) != 0;
    const unsigned long long* pin = reinterpret_cast<const unsigned long long*>(h->pin);
    BVH_HIP(hipGetLastError());
    if (best_first && (pin[7] & HEAP_OVERFLOW_BIT)) {   // a lane's heap outgrew the workspace
        if (++h->pend_attempts > 24) throw HipFail{hipErrorUnknown, "best-first heap did not converge", __LINE__};
        h->heap_cap *= 2; return false;
    }
    if (ordered && (pin[7] & 1ull)) throw HipFail{hipErrorInvalidValue, "ORDERED_DEPTH", __LINE__};
    if (h->pend_guide && (pin[7] & WALK_FLAG_GUIDE_RANGE)) {   // a ray outside the guide walk's range: this result object goes back to the f64 walk
        h->no_guide = true; h->wcounts_clean = false; return false;   // (the replay of this batch walks in f64; traverse_enqueue sets the back-off)
    }
    if (h->pend_wide && (pin[7] & 4ull)) {   // a lane's stack outgrew LDS + workspace: the binary walks need no stack
        h->force_binary = true; h->wcounts_clean = false; h->bs_clean = false; h->ckey_clean = false; h->ray_items.release(); return false;
    }
    if (flags & BVHGPU_TRAVERSE_CLOSEST) {
        if (stats) {
            h->stats.hits = pin[5];
            h->stats.device_steps = pin[1];
            h->stats.wave_steps = pin[4];
            h->stats.visited = h->pend_unfolded ? pin[1] : pin[1] + pin[5];
            h->stats.leaf_visits = h->pend_unfolded ? pin[2] : pin[5];
        }
        if (ctx->timing) ctx->ev_set |= 4u;
        h->pend_tree = nullptr;
        return true;
    }
    const unsigned long long used = pin[0];   // pool slots taken (whole chunks)
    const unsigned long long total = pin[3];  // sum of the per-ray counts = number of hits
    const bool pair_recs = h->pend_rec8;   // `used` counts 16-byte records of two hits, pool_cap 12-byte slots
    if ((pair_recs ? 2 * used : used) < total && !h->pend_staged) throw HipFail{hipErrorUnknown, "hit pool / count scan mismatch", __LINE__};
    if (total > 0xFFFFFFFFull) throw HipFail{hipErrorInvalidValue, "OVERFLOW", __LINE__};
    // Pool or index array too small: grow to the need and replay — both in ONE replay (the count scan is complete even when records were
    // dropped, and pair records need fewer pool slots than there are hits, so the pool no longer sizes the index array).
    const bool pool_small = used > (pair_recs ? (unsigned long long)h->pool_cap * sizeof(HitRec) / 16u : (unsigned long long)h->pool_cap);
    const bool idx_small = total > h->idx_cap;
    if (pool_small || idx_small) {
        if (++h->pend_attempts > 3) throw HipFail{hipErrorUnknown, pool_small ? "hit pool did not converge" : "index array did not converge", __LINE__};
        if (pool_small) {
            const size_t need = pair_recs ? ((size_t)used * 16u + sizeof(HitRec) - 1) / sizeof(HitRec) : (size_t)used;
            h->pool_cap = need + need / 8 + 1024;
        }
        if (idx_small) h->idx_cap = (size_t)total + (size_t)total / 8 + 1024;
        return false;
    }
    if (h->pend_guide) h->guide_backoff = 0;   // an f64 batch that stayed inside the guide walk's range
    h->total = total;
    h->stats.hits = total;
    if (stats) {
        h->stats.device_steps = pin[1];
        h->stats.wave_steps = pin[4];
        // reference-equivalent loop iterations (flat_bvh.rs:408): in the folded layout every
        // reported leaf stands for a navigator visit plus a leaf-entry visit
        h->stats.visited = h->pend_unfolded ? pin[1] : pin[1] + total;
        h->stats.leaf_visits = h->pend_unfolded ? pin[2] : total;
    }
    if (ctx->timing) ctx->ev_set |= 4u;
    h->pend_tree = nullptr;
    return true;
}

template <typename T>
void traverse_batch(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, unsigned flags,
                    bvhgpu_hits* h) {
    h->force_binary = false; h->pend_attempts = 0;
    for (;;) {
        traverse_enqueue<T>(t, rays_dev, n_rays, flags, h);
        BVH_HIP(hipStreamSynchronize(t->ctx->stream));
        if (traverse_check(h)) return;
    }
}

#ifdef BVH_WIDE_PROFILE
void debug_wide_prof(unsigned long long* out, size_t n) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_prof), sizeof(unsigned long long) * n);
}
void debug_wide_util(unsigned long long* out, size_t n) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_util), sizeof(unsigned long long) * n);
}
#endif

template void traverse_enqueue<float>(bvhgpu_tree*, const bvhgpu_ray_f32*, size_t, unsigned, bvhgpu_hits*);
template void traverse_enqueue<double>(bvhgpu_tree*, const bvhgpu_ray_f64*, size_t, unsigned, bvhgpu_hits*);
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
    ensure_flat_arrays(t);
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
                                                  typename Traits<T>::Ray* __restrict__ out, uint32_t stride) {
    // (a grid-stride loop: with origins / dirs in pinned HOST memory the launch is kept small — a few thousand lanes keep the PCIe link busy —
    //  so that it leaves the CUs to the build running beside it)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        // (stride 3: two arrays; stride 6 with dirs = origins + 3: origin and direction of a ray side by side)
        T o[3] = {origins[stride * (size_t)i], origins[stride * (size_t)i + 1], origins[stride * (size_t)i + 2]};
        T d[3] = {dirs[stride * (size_t)i], dirs[stride * (size_t)i + 1], dirs[stride * (size_t)i + 2]};
        ray_new<T>(o, d, out + i);
    }
}

template <typename T>
void rays_new(bvhgpu_ctx* ctx, const T* origins_dev, const T* dirs_dev, size_t n, typename Traits<T>::Ray* out_dev, hipStream_t st, unsigned max_blocks,
              unsigned stride) {
    if (!n) return;
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (max_blocks) blocks = std::min(blocks, max_blocks);
    hipLaunchKernelGGL(k_rays_new<T>, dim3(blocks), dim3(256), 0, st ? st : ctx->stream, origins_dev, dirs_dev, (uint32_t)n, out_dev, (uint32_t)stride);
    BVH_HIP(hipGetLastError());
}
template void rays_new<float>(bvhgpu_ctx*, const float*, const float*, size_t, bvhgpu_ray_f32*, hipStream_t, unsigned, unsigned);
template void rays_new<double>(bvhgpu_ctx*, const double*, const double*, size_t, bvhgpu_ray_f64*, hipStream_t, unsigned, unsigned);

// CSR offsets of one chunk of a host-resident batch (bvhgpu_traverse_host_*), moved to their place in the whole batch's array:
// out[0] holds the hits of all chunks before this one (written by the previous chunk's pass on the same stream; 0 for the first)
// (out_host: the caller's own array when it is pinned memory the device can write — the offsets then need no download)
// (first: the batch's first chunk — its base is 0 and out[0] is written here instead of read)
__global__ __launch_bounds__(256) void k_offsets_rebase(const uint32_t* __restrict__ offs, uint32_t n_plus_1, uint32_t* __restrict__ out,
                                                        uint32_t* __restrict__ out_host, uint32_t first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_plus_1) return;
    const uint32_t v = (first ? 0u : out[0]) + (i ? offs[i] : 0u);
    if (i || first) out[i] = v;        // (out[0] of a later chunk is the base itself: the previous chunk's last entry)
    if (out_host) out_host[i] = v;
}
// ... and its index list appended to the batch's (what fits into `cap` entries): base = out[0], count = offs[n_rays]
__global__ __launch_bounds__(256) void k_indices_append(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ offs, uint32_t n_rays,
                                                        const uint32_t* __restrict__ base_ptr, uint32_t* __restrict__ dst, unsigned long long cap, uint32_t first) {
    const unsigned long long base = first ? 0ull : base_ptr[0], cnt = offs[n_rays];
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < cnt && base + i < cap; i += (unsigned long long)gridDim.x * blockDim.x)
        dst[base + i] = idx[i];
}
void offsets_rebase(hipStream_t st, const uint32_t* offs_dev, size_t n_rays, uint32_t* out_dev, uint32_t* out_host, const uint32_t* idx_dev,
                    uint32_t* idx_all, size_t idx_cap, bool first) {
    // (the index list first: it reads the chunk's base out[0] and the chunk-local count, both untouched by the rebase)
    if (idx_all && idx_cap)
        hipLaunchKernelGGL(k_indices_append, dim3(128), dim3(256), 0, st, idx_dev, offs_dev, (uint32_t)n_rays, out_dev, idx_all, (unsigned long long)idx_cap,
                           first ? 1u : 0u);
    hipLaunchKernelGGL(k_offsets_rebase, dim3((unsigned)((n_rays + 1 + 255) / 256)), dim3(256), 0, st, offs_dev, (uint32_t)(n_rays + 1), out_dev, out_host,
                       first ? 1u : 0u);
    BVH_HIP(hipGetLastError());
}

// 16-byte copy (the caller's Ray structs out of pinned host memory, read by the device directly)
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void copy16(hipStream_t st, const void* src, void* dst, size_t bytes) {   // bytes: a multiple of 4; the tail goes word by word
    const size_t n16 = bytes / 16;
    if (n16) hipLaunchKernelGGL(k_copy16, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 128)), dim3(256), 0, st, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16);
    if (bytes & 15) BVH_HIP(hipMemcpyAsync(static_cast<char*>(dst) + n16 * 16, static_cast<const char*>(src) + n16 * 16, bytes & 15, hipMemcpyDefault, st));
    BVH_HIP(hipGetLastError());
}

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
