// traverse.hip — <FlatBvh as BoundingHierarchy>::traverse (src/flat_bvh.rs:396-431) for a BATCH of
// rays, plus Ray::new (src/ray/ray_impl.rs:70-80) and the bench ray stream (src/testbase.rs:687-691).
//
// One ray per lane walks the engine's folded pre-order array (common.hpp TravNode): slab test
// (src/ray/intersect_default.rs:16-37) → hit: i+1, miss: exit.  The walk visits boxes in exactly the
// reference's order, so each ray's shapes come out in the reference's (DFS, left-first) order.
// Variable-length output (Vec<&Shape> per ray) becomes CSR in three steps:
//   1. walk: every reported shape is appended to a pool as (ray, k, shape) with k = the ray's running
//      hit count; one wave-aggregated atomic per wave-iteration that has hits; counts[ray] = k_end;
//   2. exclusive scan of counts → offsets (reduce / scan-of-sums / rescan, 3 small kernels);
//   3. indices[offsets[ray] + k] = shape.
// If the pool was too small the total is still exact; the host grows it and replays.
#include "engine.hpp"

namespace bvhgpu {

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

// ctr layout (u64): [0] pool appends  [1] device steps  [2] leaf-entry steps
template <typename T, bool WITH_T, bool STATS>
__global__ __launch_bounds__(256) void k_traverse(const TravNode<T>* __restrict__ nodes, uint32_t n_trav,
                                                  const typename Traits<T>::Ray* __restrict__ rays, uint32_t n_rays,
                                                  uint32_t* __restrict__ counts, HitRec* __restrict__ pool,
                                                  T* __restrict__ pool_t, unsigned long long pool_cap,
                                                  unsigned long long* __restrict__ ctr) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    const bool active = r < n_rays;
    T o[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
    if (active) {
        const typename Traits<T>::Ray* rp = rays + r;
#pragma unroll
        for (int k = 0; k < 3; k++) { o[k] = rp->o[k]; inv[k] = rp->inv[k]; }
    }
    uint32_t i = active ? 0u : n_trav;
    uint32_t cnt = 0;
    unsigned long long steps = 0, leaf_steps = 0, wsteps = 0;
    // wave-uniform: every ray of this wave is finite → the NaN-free slab test (common.hpp) is exact
    const bool fast = !WITH_T && !__any(active && !ray_is_finite<T>(o, inv));
    while (true) {
        const bool run = i < n_trav;
        if (!__any(run)) break;
        bool rec = false;
        uint32_t shape = NONE;
        T t0 = 0, t1 = 0;
        if (STATS) wsteps++;
        if (run) {
            const NodeRegs<T> nd = load_node(nodes + i);
            const bool hit = fast ? slab_hit_finite<T>(o, inv, nd.mn, nd.mx) : slab_hit<T>(o, inv, nd.mn, nd.mx, t0, t1);
            shape = nd.shape;
            const bool leaf = trav_is_leaf(shape);
            rec = hit && leaf;
            i = hit ? i + 1 : nd.exit;   // a leaf's exit IS i+1
            if (STATS) { steps++; leaf_steps += leaf ? 1 : 0; }
        }
        const unsigned long long m = __ballot(rec);
        if (m) {
            unsigned int blo = 0, bhi = 0;
            if (lane == 0) {
                unsigned long long b = atomicAdd(&ctr[0], (unsigned long long)__popcll(m));
                blo = (unsigned int)b; bhi = (unsigned int)(b >> 32);
            }
            blo = __shfl(blo, 0); bhi = __shfl(bhi, 0);
            if (rec) {
                const unsigned long long slot = (((unsigned long long)bhi << 32) | blo) + __popcll(m & lt);
                if (slot < pool_cap) {
                    HitRec h; h.ray = r; h.k = cnt; h.shape = shape;
                    pool[slot] = h;
                    if (WITH_T) { pool_t[2 * slot] = t0; pool_t[2 * slot + 1] = t1; }
                }
                cnt++;
            }
        }
    }
    if (active) counts[r] = cnt;
    if (STATS) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            steps += __shfl_down(steps, d);
            leaf_steps += __shfl_down(leaf_steps, d);
        }
        if (lane == 0) { atomicAdd(&ctr[1], steps); atomicAdd(&ctr[2], leaf_steps); atomicAdd(&ctr[4], wsteps); }
    }
}


// ------------------------------------------------------------------------------------------------
// persistent variant: a wave owns a contiguous range of rays and keeps its 64 lanes busy — a lane whose
// ray has left the tree takes the next ray of the range (wave-uniform cursor, no atomics, no LDS).
// With one ray per lane for the whole launch a wave runs until its LONGEST ray ends (E[max of 64] is
// ~2.7x the mean walk length on the 120k-triangle scene); with refill a wave-step does useful work in
// almost every lane.  Hit records go to the pool in per-wave CHUNKs: one global atomic per 64 records
// instead of one per wave-step with a hit (hit-heavy scenes would serialise on that one address);
// the unused tail of a chunk is marked invalid (ray == NONE) for k_hits_scatter.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t POOL_CHUNK = 64;  // >= 64: one wave-step reports at most 64 hits

template <typename T, bool WITH_T, bool STATS>
__global__ __launch_bounds__(256) void k_traverse_persist(const TravNode<T>* __restrict__ nodes, uint32_t n_trav,
                                                          const typename Traits<T>::Ray* __restrict__ rays, uint32_t n_rays,
                                                          uint32_t rays_per_wave, uint32_t refill_min,
                                                          uint32_t* __restrict__ counts, HitRec* __restrict__ pool,
                                                          T* __restrict__ pool_t, unsigned long long pool_cap,
                                                          unsigned long long* __restrict__ ctr) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    const unsigned long long b0 = (unsigned long long)wave * rays_per_wave;
    const unsigned long long b1 = b0 + rays_per_wave;
    uint32_t next = (uint32_t)(b0 < n_rays ? b0 : n_rays);      // wave-uniform cursor into the range
    const uint32_t end = (uint32_t)(b1 < n_rays ? b1 : n_rays);
    next = __builtin_amdgcn_readfirstlane(next);
    T o[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
    uint32_t r = NONE, i = n_trav, cnt = 0;
    bool fin = true;               // this lane's ray has only finite components (NaN-free slab test is exact)
    unsigned long long cpos = 0;   // wave-uniform: next free slot of this wave's pool chunk
    uint32_t cleft = 0;            // wave-uniform: free slots left in it
    unsigned long long steps = 0, leaf_steps = 0, wsteps = 0;
    while (true) {
        bool run = i < n_trav;
        const unsigned long long idle = __ballot(!run);
        if (idle) {
            if (!run && r != NONE) { counts[r] = cnt; r = NONE; }   // the ray's Vec is complete
            const uint32_t nidle = (uint32_t)__popcll(idle);
            if (next < end && nidle >= refill_min) {
                const uint32_t mine = next + (uint32_t)__popcll(idle & lt);
                if (!run && mine < end) {
                    r = mine;
                    const typename Traits<T>::Ray* rp = rays + r;
#pragma unroll
                    for (int k = 0; k < 3; k++) { o[k] = rp->o[k]; inv[k] = rp->inv[k]; }
                    i = 0; cnt = 0; run = true;
                    fin = ray_is_finite<T>(o, inv);
                }
                next = (end - next) < nidle ? end : next + nidle;
            }
            if (!__any(run)) break;   // nothing in flight and the range is exhausted
        }
        bool rec = false;
        uint32_t shape = NONE;
        T t0 = 0, t1 = 0;
        if (STATS) wsteps++;
        const bool fast = !WITH_T && !__any(run && !fin);   // wave-uniform
        if (run) {
            const NodeRegs<T> nd = load_node(nodes + i);
            const bool hit = fast ? slab_hit_finite<T>(o, inv, nd.mn, nd.mx) : slab_hit<T>(o, inv, nd.mn, nd.mx, t0, t1);
            shape = nd.shape;
            const bool leaf = trav_is_leaf(shape);
            rec = hit && leaf;
            i = hit ? i + 1 : nd.exit;   // a leaf's exit IS i+1
            if (STATS) { steps++; leaf_steps += leaf ? 1 : 0; }
        }
        const unsigned long long m = __ballot(rec);
        if (m) {
            const uint32_t h = (uint32_t)__popcll(m);
            if (h > cleft) {   // wave-uniform: start a new chunk, invalidate what is left of the old one
                if ((uint32_t)lane < cleft && cpos + lane < pool_cap) pool[cpos + lane].ray = NONE;
                unsigned int blo = 0, bhi = 0;
                if (lane == 0) {
                    unsigned long long b = atomicAdd(&ctr[0], (unsigned long long)POOL_CHUNK);
                    blo = (unsigned int)b; bhi = (unsigned int)(b >> 32);
                }
                blo = __builtin_amdgcn_readfirstlane(blo); bhi = __builtin_amdgcn_readfirstlane(bhi);
                cpos = ((unsigned long long)bhi << 32) | blo;
                cleft = POOL_CHUNK;
            }
            if (rec) {
                const unsigned long long slot = cpos + __popcll(m & lt);
                if (slot < pool_cap) {
                    HitRec hr; hr.ray = r; hr.k = cnt; hr.shape = shape;
                    pool[slot] = hr;
                    if (WITH_T) { pool_t[2 * slot] = t0; pool_t[2 * slot + 1] = t1; }
                }
                cnt++;
            }
            cpos += h; cleft -= h;
        }
    }
    if ((uint32_t)lane < cleft && cpos + lane < pool_cap) pool[cpos + lane].ray = NONE;
    if (STATS) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            steps += __shfl_down(steps, d);
            leaf_steps += __shfl_down(leaf_steps, d);
        }
        if (lane == 0) { atomicAdd(&ctr[1], steps); atomicAdd(&ctr[2], leaf_steps); atomicAdd(&ctr[4], wsteps); }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-resident top of the tree.  On the 120k-triangle scene 85 % of all box tests touch the first 12
// levels of the tree (4095 entries) and the vector L1 — one tag lookup per lane per 16-byte load for
// these scattered reads — is the unit that saturates.  One 1024-thread workgroup per CU copies the
// entries whose heap number is below TopCfg<T>::SLOTS into LDS (split into 16-byte planes so that a
// ds_read_b128 of 16 lanes spreads over all 16 bank quads) and every lane tracks the slot of its
// current entry: descend → 2*slot, miss → the exit's slot carried in the entry's spare word.  A lane
// outside the resident set (deep in the tree, or after a leaf) reads HBM/L2 as before and re-enters
// the resident set through the same word.  Waves are persistent with ray refill as above.
// ------------------------------------------------------------------------------------------------
// LDS image of the resident entries: 16-byte planes of K slots each (K is a launch parameter)
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
constexpr int LDS_INNER = 8;   // walk steps between two refill phases

// The workgroup's 16 waves draw rays from ONE cursor in LDS (a wave-aggregated ds_add per refill phase),
// so the tail of a launch is the tail of a workgroup's ray range, not of every wave's.  Retiring and
// refilling lanes is kept OUT of the walk loop: LDS_INNER lean steps (~38 VALU each), then one refill
// phase; a lane whose ray ends mid-way idles for at most LDS_INNER-1 steps.
template <typename T, bool WITH_T, bool STATS>
__global__ __launch_bounds__(LDS_THREADS) void k_traverse_lds(const TravNode<T>* __restrict__ nodes, uint32_t n_trav,
                                                               const uint32_t* __restrict__ slot_entry, uint32_t K,
                                                               uint32_t first_slot,
                                                               const typename Traits<T>::Ray* __restrict__ rays,
                                                               uint32_t n_rays, uint32_t rays_per_wg,
                                                               uint32_t* __restrict__ counts, HitRec* __restrict__ pool,
                                                               T* __restrict__ pool_t, unsigned long long pool_cap,
                                                               unsigned long long* __restrict__ ctr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t& s_next = *reinterpret_cast<uint32_t*>(smem);
    TopLds<T> top(smem + 16, K);
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
    T o[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
    uint32_t r = NONE, i = n_trav, cnt = 0, slot = SLOT_NONE;
    bool fin = true;
    bool exhausted = wg_begin >= wg_end;   // wave-uniform: the workgroup's range has been handed out
    unsigned long long cpos = 0;
    uint32_t cleft = 0;
    unsigned long long steps = 0, leaf_steps = 0, wsteps = 0;
    while (true) {
        // ---- refill phase
        bool run = i < n_trav;
        const unsigned long long idle = __ballot(!run);
        if (idle) {
            if (!run && r != NONE) { counts[r] = cnt; r = NONE; }
            if (!exhausted) {
                const uint32_t nidle = (uint32_t)__popcll(idle);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_next, nidle);
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t mine = base + (uint32_t)__popcll(idle & lt);
                if (!run && base < wg_end && mine < wg_end) {
                    r = mine;
                    const typename Traits<T>::Ray* rp = rays + mine;
#pragma unroll
                    for (int k = 0; k < 3; k++) { o[k] = rp->o[k]; inv[k] = rp->inv[k]; }
                    i = 0; cnt = 0; slot = first_slot; run = true;
                    fin = ray_is_finite<T>(o, inv);
                }
                exhausted = base >= wg_end || (wg_end - base) <= nidle;
            }
            if (!__any(run)) break;
        }
        const bool fast = !WITH_T && !__any(run && !fin);   // wave-uniform
        // ---- LDS_INNER walk steps
        for (int s = 0; s < LDS_INNER; s++) {
            bool rec = false;
            uint32_t shape = NONE;
            T t0 = 0, t1 = 0;
            if (STATS) wsteps++;
            if (i < n_trav) {
                NodeRegs<T> nd;
                if (slot < K) nd = top.load(slot);
                else nd = load_node(nodes + i);
                const bool hit = fast ? slab_hit_finite<T>(o, inv, nd.mn, nd.mx) : slab_hit<T>(o, inv, nd.mn, nd.mx, t0, t1);
                shape = nd.shape;
                const bool leaf = trav_is_leaf(shape);
                rec = hit && leaf;
                const bool descend = hit && !leaf;
                i = descend ? i + 1 : nd.exit;   // a leaf's exit IS i+1
                const uint32_t child = min(slot << 1, SLOT_NONE);
                slot = descend ? child : (leaf ? SLOT_NONE : (shape & 0xFFFFu));
                if (STATS) { steps++; leaf_steps += leaf ? 1 : 0; }
            }
            const unsigned long long m = __ballot(rec);
            if (m) {
                const uint32_t h = (uint32_t)__popcll(m);
                if (h > cleft) {
                    if ((uint32_t)lane < cleft && cpos + lane < pool_cap) pool[cpos + lane].ray = NONE;
                    unsigned int blo = 0, bhi = 0;
                    if (lane == 0) {
                        unsigned long long b = atomicAdd(&ctr[0], (unsigned long long)POOL_CHUNK);
                        blo = (unsigned int)b; bhi = (unsigned int)(b >> 32);
                    }
                    blo = __builtin_amdgcn_readfirstlane(blo); bhi = __builtin_amdgcn_readfirstlane(bhi);
                    cpos = ((unsigned long long)bhi << 32) | blo;
                    cleft = POOL_CHUNK;
                }
                if (rec) {
                    const unsigned long long pslot = cpos + __popcll(m & lt);
                    if (pslot < pool_cap) {
                        HitRec hr; hr.ray = r; hr.k = cnt; hr.shape = shape;
                        pool[pslot] = hr;
                        if (WITH_T) { pool_t[2 * pslot] = t0; pool_t[2 * pslot + 1] = t1; }
                    }
                    cnt++;
                }
                cpos += h; cleft -= h;
            }
        }
    }
    if ((uint32_t)lane < cleft && cpos + lane < pool_cap) pool[cpos + lane].ray = NONE;
    if (STATS) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            steps += __shfl_down(steps, d);
            leaf_steps += __shfl_down(leaf_steps, d);
        }
        if (lane == 0) { atomicAdd(&ctr[1], steps); atomicAdd(&ctr[2], leaf_steps); atomicAdd(&ctr[4], wsteps); }
    }
}

// ---- exclusive scan of per-ray counts ----------------------------------------------------------
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;

__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ counts, uint32_t n,
                                                     unsigned long long* __restrict__ blocksums) {
    __shared__ unsigned long long ws[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) s += (base + j < n) ? counts[base + j] : 0u;
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

__global__ __launch_bounds__(256) void k_scan_final(const uint32_t* __restrict__ counts, uint32_t n,
                                                    const unsigned long long* __restrict__ blocksums,
                                                    const unsigned long long* __restrict__ total,
                                                    uint32_t* __restrict__ offsets) {
    __shared__ uint32_t ws[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) { v[j] = (base + j < n) ? counts[base + j] : 0u; s += v[j]; }
    uint32_t inc = s;
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t u = __shfl_up(inc, d);
        if (lane >= d) inc += u;
    }
    if (lane == WAVE - 1) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) wbase += ws[w];
    uint32_t run = (uint32_t)blocksums[blockIdx.x] + wbase + inc - s;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        if (base + j < n) offsets[base + j] = run;
        run += v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n] = (uint32_t)(*total);
}

template <typename T, bool WITH_T>
__global__ __launch_bounds__(256) void k_hits_scatter(const HitRec* __restrict__ pool, const T* __restrict__ pool_t,
                                                      const unsigned long long* __restrict__ ctr,
                                                      unsigned long long pool_cap, const uint32_t* __restrict__ offsets,
                                                      uint32_t* __restrict__ indices, T* __restrict__ tslice) {
    const unsigned long long n = ctr[0];
    if (n > pool_cap) return;  // pool overflowed: indices[] is too small as well; the host grows both and replays
    for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < n;
         j += (unsigned long long)gridDim.x * blockDim.x) {
        const HitRec h = pool[j];
        if (h.ray == NONE) continue;   // unused tail of a per-wave chunk
        const uint32_t d = offsets[h.ray] + h.k;
        indices[d] = h.shape;
        if (WITH_T) { tslice[2 * (size_t)d] = pool_t[2 * j]; tslice[2 * (size_t)d + 1] = pool_t[2 * j + 1]; }
    }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
void traverse_batch(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, unsigned flags,
                    bvhgpu_hits* h) {
    bvhgpu_ctx* ctx = t->ctx;
    hipStream_t st = ctx->stream;
    const bool with_t = (flags & BVHGPU_TRAVERSE_T_SLICE) != 0;
    const bool stats = (flags & BVHGPU_TRAVERSE_STATS) != 0;
    h->ctx = ctx; h->dtype = Traits<T>::dtype; h->n_rays = n_rays; h->flags = flags; h->total = 0;
    h->stats = bvhgpu_traverse_stats{0, 0, 0, 0, 0};
    h->counts.reserve((n_rays + 1) * 4);
    h->offsets.reserve((n_rays + 1) * 4);
    h->ctr.reserve(8 * sizeof(unsigned long long));
    const uint32_t nb = (uint32_t)((n_rays + SCAN_BLOCK - 1) / SCAN_BLOCK);
    h->blocksums.reserve((nb + 1) * sizeof(unsigned long long));
    if (h->pool_cap == 0) h->pool_cap = std::max<size_t>(n_rays, (size_t)1 << 16);
    unsigned long long* pin = reinterpret_cast<unsigned long long*>(ctx->pinned);

    if (n_rays == 0) {
        BVH_HIP(hipMemsetAsync(h->offsets.p, 0, 4, st));
        BVH_HIP(hipStreamSynchronize(st));
        return;
    }
    for (int attempt = 0; attempt < 2; attempt++) {
        h->pool.reserve(h->pool_cap * sizeof(HitRec));
        h->indices.reserve(h->pool_cap * 4);
        if (with_t) {
            h->pool_t.reserve(h->pool_cap * 2 * sizeof(T));
            h->tslice.reserve(h->pool_cap * 2 * sizeof(T));
        }
        unsigned long long* ctr = h->ctr.as<unsigned long long>();
        BVH_HIP(hipMemsetAsync(ctr, 0, 8 * sizeof(unsigned long long), st));
        const uint32_t n_trav = (uint32_t)t->n_trav;
        const dim3 grid((unsigned)((n_rays + 255) / 256)), block(256);
        const TravNode<T>* nodes = t->trav.as<TravNode<T>>();
        uint32_t* counts = h->counts.as<uint32_t>();
        HitRec* pool = h->pool.as<HitRec>();
        T* pool_t = h->pool_t.as<T>();
        const unsigned long long cap = h->pool_cap;
        if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[4], st)); }
// launch geometry.  variant 0: one ray per lane per launch; 1: persistent waves with ray refill;
        // 2: persistent + LDS-resident top of the tree (one 1024-thread workgroup per CU)
        int variant = ctx->tune[BVHGPU_TUNE_TRAVERSE_VARIANT];
        if (variant == 2 && (t->slot_entry.p == nullptr || n_rays < (size_t)ctx->tune[BVHGPU_TUNE_TRAVERSE_LDS_MIN_RAYS]))
            variant = 0;
        const bool persist = variant != 0;
        const size_t full = (n_rays + WAVE - 1) / WAVE;
        // variant 2 geometry: workgroups of lds_threads that each keep K top-of-tree slots in LDS; as many
        // workgroups per CU as 160 KB of LDS and 32 waves allow
        const uint32_t lds_threads = (uint32_t)std::min(LDS_THREADS, std::max(64, ctx->tune[BVHGPU_TUNE_TRAVERSE_LDS_THREADS] & ~63));
        const uint32_t K = (uint32_t)std::min<int>((int)TopCfg<T>::SLOTS, std::max(4, ctx->tune[BVHGPU_TUNE_TRAVERSE_LDS_SLOTS]));
        const size_t lds_bytes = 16 + (size_t)K * TopLds<T>::BYTES_PER_SLOT;
        const uint32_t wg_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds_bytes, 2048 / lds_threads));
        const uint32_t wpc = variant == 2 ? wg_per_cu * lds_threads / WAVE
                                          : (uint32_t)std::max(1, ctx->tune[BVHGPU_TUNE_TRAVERSE_WAVES_PER_CU]);
        const uint32_t n_waves = (uint32_t)std::min<size_t>(full, (size_t)ctx->n_cu * wpc);
        const uint32_t rpw = (uint32_t)((n_rays + n_waves - 1) / n_waves);
        const uint32_t refill_min = (uint32_t)std::min(64, std::max(1, ctx->tune[BVHGPU_TUNE_TRAVERSE_REFILL_MIN]));
        const dim3 pgrid((n_waves + 3) / 4);
        const dim3 lgrid((n_waves + lds_threads / WAVE - 1) / (lds_threads / WAVE));
        const uint32_t first_slot = t->n >= 2 ? 2u : SLOT_NONE;   // entry 0 is the root's left child (heap number 2)
        const uint32_t* slot_entry = t->slot_entry.as<uint32_t>();
        const uint32_t rpg = (uint32_t)((n_rays + lgrid.x - 1) / lgrid.x);   // rays per workgroup (variant 2)
#define LAUNCH_TRAV(WT, STT)                                                                                               \
    do {                                                                                                                   \
        if (variant == 2)                                                                                                  \
        {                                                                                                                  \
            BVH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_traverse_lds<T, WT, STT>),                        \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                      \
            hipLaunchKernelGGL((k_traverse_lds<T, WT, STT>), lgrid, dim3(lds_threads), lds_bytes, st, nodes, n_trav,       \
                               slot_entry, K, first_slot, rays_dev, (uint32_t)n_rays, rpg, counts, pool, pool_t, cap, ctr);\
        }                                                                                                                  \
        else if (variant == 1)                                                                                             \
            hipLaunchKernelGGL((k_traverse_persist<T, WT, STT>), pgrid, block, 0, st, nodes, n_trav, rays_dev,             \
                               (uint32_t)n_rays, rpw, refill_min, counts, pool, pool_t, cap, ctr);                         \
        else                                                                                                               \
            hipLaunchKernelGGL((k_traverse<T, WT, STT>), grid, block, 0, st, nodes, n_trav, rays_dev, (uint32_t)n_rays,    \
                               counts, pool, pool_t, cap, ctr);                                                            \
    } while (0)
        if (with_t) { if (stats) LAUNCH_TRAV(true, true); else LAUNCH_TRAV(true, false); }
        else { if (stats) LAUNCH_TRAV(false, true); else LAUNCH_TRAV(false, false); }
#undef LAUNCH_TRAV
        if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[5], st)); }
        unsigned long long* bs = h->blocksums.as<unsigned long long>();
        hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(256), 0, st, counts, (uint32_t)n_rays, bs);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, bs, nb, ctr + 3);
        hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(256), 0, st, counts, (uint32_t)n_rays, bs, ctr + 3,
                           h->offsets.as<uint32_t>());
        const int sgrid = (int)std::min<size_t>((cap + 255) / 256, (size_t)ctx->n_cu * 8);
        if (with_t)
            hipLaunchKernelGGL((k_hits_scatter<T, true>), dim3(sgrid), dim3(256), 0, st, pool, pool_t, ctr, cap,
                               h->offsets.as<uint32_t>(), h->indices.as<uint32_t>(), h->tslice.as<T>());
        else
            hipLaunchKernelGGL((k_hits_scatter<T, false>), dim3(sgrid), dim3(256), 0, st, pool, pool_t, ctr, cap,
                               h->offsets.as<uint32_t>(), h->indices.as<uint32_t>(), h->tslice.as<T>());
        if (ctx->timing) { BVH_HIP(hipEventRecord(ctx->ev[6], st)); }
        BVH_HIP(hipMemcpyAsync(pin, ctr, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        BVH_HIP(hipStreamSynchronize(st));
        BVH_HIP(hipGetLastError());
        const unsigned long long used = pin[0];   // pool slots taken (persistent variant: whole chunks)
        const unsigned long long total = pin[3];  // sum of the per-ray counts = number of hits
        if (used < total || (!persist && used != total)) throw HipFail{hipErrorUnknown, "hit pool / count scan mismatch", __LINE__};
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
}

template void traverse_batch<float>(bvhgpu_tree*, const bvhgpu_ray_f32*, size_t, unsigned, bvhgpu_hits*);
template void traverse_batch<double>(bvhgpu_tree*, const bvhgpu_ray_f64*, size_t, unsigned, bvhgpu_hits*);

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
