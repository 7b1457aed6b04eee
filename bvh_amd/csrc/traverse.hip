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
    unsigned long long steps = 0, leaf_steps = 0;
    while (true) {
        const bool run = i < n_trav;
        if (!__any(run)) break;
        bool rec = false;
        uint32_t shape = NONE;
        T t0 = 0, t1 = 0;
        if (run) {
            const NodeRegs<T> nd = load_node(nodes + i);
            const bool hit = slab_hit<T>(o, inv, nd.mn, nd.mx, t0, t1);
            shape = nd.shape;
            rec = hit && (shape != NONE);
            i = hit ? i + 1 : nd.exit;
            if (STATS) { steps++; leaf_steps += (shape != NONE) ? 1 : 0; }
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
        if (lane == 0) { atomicAdd(&ctr[1], steps); atomicAdd(&ctr[2], leaf_steps); }
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
    h->stats = bvhgpu_traverse_stats{0, 0, 0, 0};
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
#define LAUNCH_TRAV(WT, STT) \
    hipLaunchKernelGGL((k_traverse<T, WT, STT>), grid, block, 0, st, nodes, n_trav, rays_dev, (uint32_t)n_rays, counts, \
                       pool, pool_t, cap, ctr)
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
        const unsigned long long total = pin[0];
        if (total != pin[3]) throw HipFail{hipErrorUnknown, "hit pool / count scan mismatch", __LINE__};
        if (total > 0xFFFFFFFFull) throw HipFail{hipErrorInvalidValue, "OVERFLOW", __LINE__};
        if (total > cap) {  // pool too small: grow to the exact need and replay (deterministic)
            h->pool_cap = (size_t)total + (size_t)total / 8 + 1024;
            continue;
        }
        h->total = total;
        h->stats.hits = total;
        if (stats) {
            h->stats.device_steps = pin[1];
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
