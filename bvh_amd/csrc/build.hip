// build.hip — top-down 6-bucket SAH builder of the reference (src/bvh/bvh_node.rs:81-279,
// src/bvh/bvh_impl.rs:53-96, src/utils.rs:59-109), re-designed for MI355X and bit-exact with it.
//
// The reference recurses node by node (rayon::join).  Node placement is arithmetic (pre-order:
// left = ni+1, right = ni+1+(2*nl-1), bvh_node.rs:138-142), so any node can be split as soon as
// its index slice is final, in any order.  Two tiers:
//
//  tier 1 — level-synchronous, for nodes with more than 64 shapes.  Per level three kernels over a
//           queue of work items (one item = one BvhNodeBuildArgs):
//             bin     : one workgroup per 1024-position tile of an item: bucket id per shape
//                       (bvh_node.rs:204-222), 6 x (count, AABB, centroid-AABB) reduced with LDS
//                       integer-key atomics, merged into the item's stats with global atomics
//                       (min/max are exact → order-free);
//             select  : one wave per item: 5 candidate splits, strict-< first-wins argmin
//                       (:231-247), writes the BvhNode, creates the child items, turns per-tile
//                       bucket counts into exclusive scatter offsets;
//             scatter : stable bucket-major rewrite of the index slice (:250-272) = one stable
//                       3-bit counting-sort pass: wave ballot ranks + per-tile offsets.
//  tier 2 — wave-subtree: every node with <= 64 shapes is finished by ONE wavefront, one shape per
//           lane, all levels of the subtree at once: segmented ballot ranks for the stable sort,
//           ds_permute to move shapes, segmented min/max prefix+suffix scans for the L/R bounds
//           of the 5 candidate splits (plays the role of rayon_executor's sequential cut-off,
//           bvh_impl.rs:534).
#include "engine.hpp"

namespace bvhgpu {

constexpr int MAXLV = 96;        // counter slots (levels beyond reuse the last two, host-synchronised)
constexpr int CTR_SMALL = 0;     // u32: number of small items
constexpr int CTR_LEVEL0 = 16;   // u32 pairs (n_items, n_tiles) per level slot
constexpr size_t ROOTKEY_OFF = 1024;  // byte offset of the 12 root keys inside the ctr buffer

__host__ __device__ inline int lvl_slot(int L) { return L < MAXLV - 2 ? L : (MAXLV - 2 + (L & 1)); }

template <typename T> struct ItemStats {
    typename Traits<T>::Key k[NUM_BUCKETS * STAT_KEYS];
    uint32_t cnt[NUM_BUCKETS];
    uint32_t _pad[2];
};

template <typename T> struct BuildArgs {
    const T* aabbs;
    typename Traits<T>::Node* nodes;
    uint32_t* node_start;
    uint32_t* node_count;
    uint32_t* shape_node;
    uint32_t* idx[2];
    uint8_t* bk;
    Item<T>* big[2];
    Item<T>* small;
    ItemStats<T>* stats[2];
    uint32_t* tile_item[2];
    uint32_t* tile_cnt;
    uint32_t* ctr;
    typename Traits<T>::Key* rootkeys;
    uint32_t n;
};

// is key slot j of a bucket a "min" slot?  layout: aabb.min[3] aabb.max[3] cen.min[3] cen.max[3]
__device__ __forceinline__ bool key_is_min(int j) { return j < 3 || (j >= 6 && j < 9); }

// ------------------------------------------------------------------------------------------------
// K1 prep: identity permutation (bvh_impl.rs:61-63) + joint_aabb_of_shapes over all shapes (:74)
// ------------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void k_prep(BuildArgs<T> a) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    __shared__ Key sk[STAT_KEYS];
    if (threadIdx.x < STAT_KEYS) sk[threadIdx.x] = key_is_min(threadIdx.x) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
    __syncthreads();
    Key loc[STAT_KEYS];
#pragma unroll
    for (int j = 0; j < STAT_KEYS; j++) loc[j] = key_is_min(j) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
        a.idx[0][i] = i;
        const T* b = a.aabbs + 6 * (size_t)i;
        T bx[6];
#pragma unroll
        for (int k = 0; k < 6; k++) bx[k] = b[k];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            Key kmn = Tr::key(bx[k]), kmx = Tr::key(bx[3 + k]);
            Key kc = Tr::key(center1(bx[k], bx[3 + k]));
            loc[k] = loc[k] < kmn ? loc[k] : kmn;
            loc[3 + k] = loc[3 + k] > kmx ? loc[3 + k] : kmx;
            loc[6 + k] = loc[6 + k] < kc ? loc[6 + k] : kc;
            loc[9 + k] = loc[9 + k] > kc ? loc[9 + k] : kc;
        }
    }
#pragma unroll
    for (int j = 0; j < STAT_KEYS; j++) {
        if (key_is_min(j)) atomicMin(&sk[j], loc[j]);
        else atomicMax(&sk[j], loc[j]);
    }
    __syncthreads();
    if (threadIdx.x < STAT_KEYS) {
        int j = threadIdx.x;
        if (key_is_min(j)) atomicMin(&a.rootkeys[j], sk[j]);
        else atomicMax(&a.rootkeys[j], sk[j]);
    }
}

template <typename T> __device__ void init_stats(ItemStats<T>* s, int lane) {
    using Tr = Traits<T>;
    for (int j = lane; j < NUM_BUCKETS * STAT_KEYS; j += WAVE)
        s->k[j] = key_is_min(j % STAT_KEYS) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
    if (lane < NUM_BUCKETS) s->cnt[lane] = 0;
}

// enqueue one child / root work item (whole wave participates; lane 0 owns the atomics)
template <typename T>
__device__ void push_item(const BuildArgs<T>& a, int next_level, uint32_t ni, uint32_t parent, uint32_t start,
                          uint32_t count, const T* A, const T* C, int lane) {
    const int nslot = lvl_slot(next_level);
    const int npar = next_level & 1;
    uint32_t slot = 0, tb = 0;
    const bool is_small = count <= (uint32_t)SMALL_MAX;
    const uint32_t ntile = (count + TILE - 1) / TILE;
    if (lane == 0) {
        if (is_small) slot = atomicAdd(&a.ctr[CTR_SMALL], 1u);
        else {
            slot = atomicAdd(&a.ctr[CTR_LEVEL0 + 2 * nslot], 1u);
            tb = atomicAdd(&a.ctr[CTR_LEVEL0 + 2 * nslot + 1], ntile);
        }
    }
    slot = __shfl(slot, 0);
    tb = __shfl(tb, 0);
    Item<T>* it = is_small ? &a.small[slot] : &a.big[npar][slot];
    if (lane == 0) {
        it->ni = ni; it->parent = parent; it->start = start; it->count = count;
        it->tile_base = tb; it->parity = (uint32_t)npar; it->_r0 = 0; it->_r1 = 0;
    }
    if (lane < 6) { it->A[lane] = A[lane]; it->C[lane] = C[lane]; }
    if (!is_small) {
        for (uint32_t j = lane; j < ntile; j += WAVE) a.tile_item[npar][tb + j] = slot;
        init_stats<T>(&a.stats[npar][slot], lane);
    }
}

// counters = 0, root keys = identities of min / max
template <typename T> __global__ __launch_bounds__(256) void k_init(BuildArgs<T> a) {
    using Tr = Traits<T>;
    for (int i = threadIdx.x; i < (int)(ROOTKEY_OFF / 4); i += 256) a.ctr[i] = 0;
    if (threadIdx.x < STAT_KEYS) a.rootkeys[threadIdx.x] = key_is_min(threadIdx.x) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
}

// root item: BvhNodeBuildArgs of bvh_impl.rs:75-87
template <typename T> __global__ __launch_bounds__(64) void k_root(BuildArgs<T> a) {
    using Tr = Traits<T>;
    const int lane = lane_id();
    T A[6], C[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        A[k] = Tr::unkey(a.rootkeys[k]);
        C[k] = Tr::unkey(a.rootkeys[6 + k]);
    }
    push_item<T>(a, 0, 0u, 0u, 0u, a.n, A, C, lane);
}

// ------------------------------------------------------------------------------------------------
// tier 1 / bin
// ------------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void k_bin(BuildArgs<T> a, int level) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    const int slot = lvl_slot(level), par = level & 1;
    const uint32_t ntiles = a.ctr[CTR_LEVEL0 + 2 * slot + 1];
    __shared__ Key sk[NUM_BUCKETS * STAT_KEYS];
    __shared__ uint32_t sc[NUM_BUCKETS];
    const uint32_t* idx = a.idx[par];
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t item_id = a.tile_item[par][t];
        const Item<T>* it = &a.big[par][item_id];
        const uint32_t start = it->start, count = it->count;
        const uint32_t p0 = start + (t - it->tile_base) * TILE;
        const uint32_t pend = min(start + count, p0 + (uint32_t)TILE);
        T C[6];
#pragma unroll
        for (int k = 0; k < 6; k++) C[k] = it->C[k];
        const int ax = largest_axis(C);                 // bvh_node.rs:107
        const T cmin = C[ax];
        const T ext = C[3 + ax] - C[ax];                // :108
        const bool degen = ext < Tr::eps();             // :114
        const uint32_t half = count / 2;                // :117
        for (int j = threadIdx.x; j < NUM_BUCKETS * STAT_KEYS; j += 256)
            sk[j] = key_is_min(j % STAT_KEYS) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
        if (threadIdx.x < NUM_BUCKETS) sc[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t p = p0 + threadIdx.x; p < pend; p += 256) {
            const uint32_t s = idx[p];
            const T* b = a.aabbs + 6 * (size_t)s;
            T bx[6];
#pragma unroll
            for (int k = 0; k < 6; k++) bx[k] = b[k];
            T c[3];
#pragma unroll
            for (int k = 0; k < 3; k++) c[k] = center1(bx[k], bx[3 + k]);
            int bkt;
            if (degen) bkt = (p - start) < half ? 0 : 1;   // halves in CURRENT order (:117)
            else bkt = bucket_of(c[ax], cmin, ext);        // :210-217
            a.bk[p] = (uint8_t)bkt;
            Key* kk = sk + bkt * STAT_KEYS;                // Bucket::add_aabb (utils.rs:81-85)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                atomicMin(&kk[k], Tr::key(bx[k]));
                atomicMax(&kk[3 + k], Tr::key(bx[3 + k]));
                Key kc = Tr::key(c[k]);
                atomicMin(&kk[6 + k], kc);
                atomicMax(&kk[9 + k], kc);
            }
            atomicAdd(&sc[bkt], 1u);
        }
        __syncthreads();
        ItemStats<T>* gs = &a.stats[par][item_id];
        if (threadIdx.x < NUM_BUCKETS) {
            a.tile_cnt[t * NUM_BUCKETS + threadIdx.x] = sc[threadIdx.x];
            if (sc[threadIdx.x]) atomicAdd(&gs->cnt[threadIdx.x], sc[threadIdx.x]);
        }
        if (threadIdx.x < NUM_BUCKETS * STAT_KEYS) {
            const int j = threadIdx.x;
            if (sc[j / STAT_KEYS]) {
                if (key_is_min(j % STAT_KEYS)) atomicMin(&gs->k[j], sk[j]);
                else atomicMax(&gs->k[j], sk[j]);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// tier 1 / select — one wave per item
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void box_empty(T* b) {
    b[0] = b[1] = b[2] = Traits<T>::inf();
    b[3] = b[4] = b[5] = -Traits<T>::inf();
}
template <typename T> __device__ __forceinline__ void box_join(T* a, const T* b) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        a[k] = tmin(a[k], b[k]);
        a[3 + k] = tmax(a[3 + k], b[3 + k]);
    }
}

template <typename T> __global__ __launch_bounds__(256) void k_select(BuildArgs<T> a, int level) {
    using Tr = Traits<T>;
    const int slot = lvl_slot(level), par = level & 1;
    const uint32_t nitems = a.ctr[CTR_LEVEL0 + 2 * slot];
    const int lane = lane_id();
    const uint32_t wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t id = wave0; id < nitems; id += nwaves) {
        const Item<T>* it = &a.big[par][id];
        const ItemStats<T>* st = &a.stats[par][id];
        const uint32_t ni = it->ni, parent = it->parent, start = it->start, count = it->count;
        T A[6], C[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { A[k] = it->A[k]; C[k] = it->C[k]; }
        const int ax = largest_axis(C);
        const T ext = C[3 + ax] - C[ax];
        const bool degen = ext < Tr::eps();

        uint32_t cnt[NUM_BUCKETS];
        T ba[NUM_BUCKETS][6], bc[NUM_BUCKETS][6];
#pragma unroll
        for (int b = 0; b < NUM_BUCKETS; b++) {
            cnt[b] = st->cnt[b];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                ba[b][k] = Tr::unkey(st->k[b * STAT_KEYS + k]);
                bc[b][k] = Tr::unkey(st->k[b * STAT_KEYS + 6 + k]);
            }
        }
        // candidate splits (bvh_node.rs:224-247); in the degenerate branch (:114-124) "bucket" 0/1 are
        // the two halves and the split is forced between them (joint_aabb_of_shapes of each half)
        int best = 0;
        T min_cost = Tr::inf();
        T AL[6], CL[6], AR[6], CR[6];
        box_empty(AL); box_empty(CL); box_empty(AR); box_empty(CR);
        // if no candidate ever wins (NaN/inf costs) the reference keeps min_bucket = 0 and EMPTY child
        // bounds (bvh_node.rs:225-230,250): replicate
        uint32_t nl = cnt[0];
        const T sa_parent = surface_area(A);
#pragma unroll
        for (int s = 0; s < NUM_BUCKETS - 1; s++) {
            uint32_t ln = 0, rn = 0;
            T la[6], lc[6], ra[6], rc[6];
            box_empty(la); box_empty(lc); box_empty(ra); box_empty(rc);
#pragma unroll
            for (int b = 0; b < NUM_BUCKETS; b++) {
                if (b <= s) { ln += cnt[b]; box_join(la, ba[b]); box_join(lc, bc[b]); }
                else { rn += cnt[b]; box_join(ra, ba[b]); box_join(rc, bc[b]); }
            }
            T cl = (T)ln * surface_area(la);
            T cr = (T)rn * surface_area(ra);
            T num = cl + cr;
            T cost = num / sa_parent;
            bool take = degen ? (s == 0) : (cost < min_cost);
            if (take) {
                best = s; min_cost = cost; nl = ln;
#pragma unroll
                for (int k = 0; k < 6; k++) { AL[k] = la[k]; CL[k] = lc[k]; AR[k] = ra[k]; CR[k] = rc[k]; }
            }
        }
        (void)best;
        const uint32_t li = ni + 1;                 // :140
        const uint32_t ri = li + (2 * nl - 1);      // :138,142
        if (lane == 0) {
            typename Tr::Node* nd = &a.nodes[ni];   // :145-151
#pragma unroll
            for (int k = 0; k < 3; k++) {
                nd->l_min[k] = AL[k]; nd->l_max[k] = AL[3 + k];
                nd->r_min[k] = AR[k]; nd->r_max[k] = AR[3 + k];
            }
            nd->parent = parent; nd->l = li; nd->r = ri; nd->shape = NONE;
            a.node_start[ni] = start;
            a.node_count[ni] = count;
        }
        push_item<T>(a, level + 1, li, ni, start, nl, AL, CL, lane);
        push_item<T>(a, level + 1, ri, ni, start + nl, count - nl, AR, CR, lane);

        // per-tile exclusive offsets for the stable scatter: dest = start + tile_cnt[t][b] + rank
        uint32_t base[NUM_BUCKETS];
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < NUM_BUCKETS; b++) { base[b] = acc; acc += cnt[b]; }
        const uint32_t ntile = (count + TILE - 1) / TILE;
        for (uint32_t c0 = 0; c0 < ntile; c0 += WAVE) {
            const uint32_t tl = c0 + lane;
            const bool valid = tl < ntile;
            uint32_t* tc = a.tile_cnt + (size_t)(it->tile_base + tl) * NUM_BUCKETS;
#pragma unroll
            for (int b = 0; b < NUM_BUCKETS; b++) {
                uint32_t v = valid ? tc[b] : 0u;
                uint32_t inc = v;
#pragma unroll
                for (int d = 1; d < WAVE; d <<= 1) {
                    uint32_t u = __shfl_up(inc, d);
                    if (lane >= d) inc += u;
                }
                if (valid) tc[b] = base[b] + inc - v;
                base[b] += __shfl(inc, WAVE - 1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// tier 1 / scatter — stable bucket-major rewrite (bvh_node.rs:250-272)
// ------------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void k_scatter(BuildArgs<T> a, int level) {
    const int slot = lvl_slot(level), par = level & 1;
    const uint32_t ntiles = a.ctr[CTR_LEVEL0 + 2 * slot + 1];
    __shared__ uint32_t run[NUM_BUCKETS];
    __shared__ uint32_t wcnt[4][NUM_BUCKETS];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const unsigned long long lt = lanemask_lt();
    const uint32_t* src = a.idx[par];
    uint32_t* dst = a.idx[par ^ 1];
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const Item<T>* it = &a.big[par][a.tile_item[par][t]];
        const uint32_t start = it->start, count = it->count;
        const uint32_t p0 = start + (t - it->tile_base) * TILE;
        const uint32_t pend = min(start + count, p0 + (uint32_t)TILE);
        if (threadIdx.x < NUM_BUCKETS) run[threadIdx.x] = a.tile_cnt[(size_t)t * NUM_BUCKETS + threadIdx.x];
        __syncthreads();
        for (uint32_t c0 = p0; c0 < pend; c0 += 256) {
            const uint32_t p = c0 + threadIdx.x;
            const bool valid = p < pend;
            const int b = valid ? (int)a.bk[p] : 7;
            const uint32_t s = valid ? src[p] : 0u;
            uint32_t rank = 0;
#pragma unroll
            for (int bb = 0; bb < NUM_BUCKETS; bb++) {
                unsigned long long m = __ballot(b == bb);
                if (b == bb) rank = (uint32_t)__popcll(m & lt);
                if (lane == 0) wcnt[w][bb] = (uint32_t)__popcll(m);
            }
            __syncthreads();
            if (valid) {
                uint32_t off = run[b] + rank;
                for (int ww = 0; ww < w; ww++) off += wcnt[ww][b];
                dst[start + off] = s;
            }
            __syncthreads();
            if (threadIdx.x < NUM_BUCKETS)
                run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// tier 2 — wave-subtree kernel
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float push_lane(float v, int dst) {
    return __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(v)));
}
__device__ __forceinline__ uint32_t push_lane(uint32_t v, int dst) {
    return (uint32_t)__builtin_amdgcn_ds_permute(dst << 2, (int)v);
}
__device__ __forceinline__ double push_lane(double v, int dst) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_ds_permute(dst << 2, (int)(b & 0xFFFFFFFFll));
    int hi = __builtin_amdgcn_ds_permute(dst << 2, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <typename T> __global__ __launch_bounds__(256) void k_small(BuildArgs<T> a, uint32_t n_small) {
    using Tr = Traits<T>;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= n_small) return;
    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    const Item<T>* it = &a.small[wave];
    const uint32_t istart = it->start;
    const int n = (int)it->count;
    const uint32_t* idx = a.idx[it->parity];

    bool done = lane >= n;
    uint32_t shape = done ? 0u : idx[istart + lane];
    T box[6];
    if (!done) {
        const T* b = a.aabbs + 6 * (size_t)shape;
#pragma unroll
        for (int k = 0; k < 6; k++) box[k] = b[k];
    } else {
        box_empty(box);
    }
    // per-lane copy of the segment (= BvhNodeBuildArgs) the lane currently belongs to
    int lo = done ? lane : 0, hi = done ? lane + 1 : n;
    uint32_t ni = it->ni, parent = it->parent;
    T Cb[6], A0[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { Cb[k] = it->C[k]; A0[k] = it->A[k]; }
    T saA = surface_area(A0);

    while (true) {
        int segn = hi - lo;
        if (!done && segn == 1) {  // bvh_node.rs:95-104
            typename Tr::Node* nd = &a.nodes[ni];
#pragma unroll
            for (int k = 0; k < 3; k++) { nd->l_min[k] = 0; nd->l_max[k] = 0; nd->r_min[k] = 0; nd->r_max[k] = 0; }
            nd->parent = parent; nd->l = NONE; nd->r = NONE; nd->shape = shape;
            a.shape_node[shape] = ni;       // set_bh_node_index (:102)
            a.node_start[ni] = istart + lane;
            a.node_count[ni] = 1;
            done = true;
            lo = lane; hi = lane + 1;
        }
        if (__ballot(!done) == 0ull) break;

        // ---- bucket id (bvh_node.rs:107-108, 114-117, 204-217)
        T c[3];
#pragma unroll
        for (int k = 0; k < 3; k++) c[k] = center1(box[k], box[3 + k]);
        const int ax = largest_axis(Cb);
        const T cmin = ax == 0 ? Cb[0] : (ax == 1 ? Cb[1] : Cb[2]);
        const T cmax = ax == 0 ? Cb[3] : (ax == 1 ? Cb[4] : Cb[5]);
        const T cax = ax == 0 ? c[0] : (ax == 1 ? c[1] : c[2]);
        const T ext = cmax - cmin;
        const bool degen = ext < Tr::eps();
        int b = 7;
        if (!done) b = degen ? ((lane - lo) < segn / 2 ? 0 : 1) : bucket_of(cax, cmin, ext);

        // ---- stable bucket-major position inside the segment (:250-272)
        const unsigned long long segmask = done ? 0ull : mask_range(lo, hi);
        int cnt[NUM_BUCKETS];
        int basec = 0, rank = 0;
#pragma unroll
        for (int bb = 0; bb < NUM_BUCKETS; bb++) {
            const unsigned long long m = __ballot(b == bb) & segmask;
            cnt[bb] = __popcll(m);
            if (bb < b) basec += cnt[bb];
            if (bb == b) rank = __popcll(m & lt);
        }
        const int dstl = done ? lane : lo + basec + rank;
        shape = push_lane(shape, dstl);
#pragma unroll
        for (int k = 0; k < 6; k++) box[k] = push_lane(box[k], dstl);
#pragma unroll
        for (int k = 0; k < 3; k++) c[k] = center1(box[k], box[3 + k]);

        // ---- segmented inclusive prefix (P) and suffix (S) joins of AABB and centroid bounds
        T P[12], S[12];
#pragma unroll
        for (int k = 0; k < 6; k++) { P[k] = box[k]; S[k] = box[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { P[6 + k] = c[k]; P[9 + k] = c[k]; S[6 + k] = c[k]; S[9 + k] = c[k]; }
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const bool up_ok = (lane - d) >= lo;
            const bool dn_ok = (lane + d) < hi;
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const bool is_min = key_is_min(k);
                T u = __shfl_up(P[k], d);
                T v = __shfl_down(S[k], d);
                if (up_ok) P[k] = is_min ? tmin(P[k], u) : tmax(P[k], u);
                if (dn_ok) S[k] = is_min ? tmin(S[k], v) : tmax(S[k], v);
            }
        }
        // ---- SAH cost of the 5 candidates (:231-247): L = prefix at the last lane of bucket <= s,
        //      R = suffix at the first lane of bucket > s
        const T saP = surface_area(P), saS = surface_area(S);
        int nl = cnt[0];
        bool taken = false;  // no winner (NaN/inf costs) → min_bucket 0 with EMPTY child bounds (:225-230)
        T min_cost = Tr::inf();
        int cum = 0;
#pragma unroll
        for (int s = 0; s < NUM_BUCKETS - 1; s++) {
            cum += cnt[s];
            int q = lo + cum;
            int ql = min(max(q - 1, 0), 63), qr = min(max(q, 0), 63);
            T sal = __shfl(saP, ql);
            T sar = __shfl(saS, qr);
            T cl = (T)cum * sal;
            T cr = (T)(segn - cum) * sar;
            T num = cl + cr;
            T cost = num / saA;
            bool take = degen ? (s == 0) : (cost < min_cost);
            if (take) { min_cost = cost; nl = cum; taken = true; }
        }
        const int q = lo + nl;
        const int ql = min(max(q - 1, 0), 63), qr = min(max(q, 0), 63);
        T AL[6], AR[6], Cn[6];
        const bool left = lane < q;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            AL[k] = __shfl(P[k], ql);
            AR[k] = __shfl(S[k], qr);
            T cl = __shfl(P[6 + k], ql);
            T cr = __shfl(S[6 + k], qr);
            Cn[k] = left ? cl : cr;
        }
        if (!taken) { box_empty(AL); box_empty(AR); box_empty(Cn); }
        if (!done) {
            const uint32_t li = ni + 1;
            const uint32_t ri = li + (uint32_t)(2 * nl - 1);
            if (lane == lo) {  // bvh_node.rs:145-151
                typename Tr::Node* nd = &a.nodes[ni];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    nd->l_min[k] = AL[k]; nd->l_max[k] = AL[3 + k];
                    nd->r_min[k] = AR[k]; nd->r_max[k] = AR[3 + k];
                }
                nd->parent = parent; nd->l = li; nd->r = ri; nd->shape = NONE;
                a.node_start[ni] = istart + lo;
                a.node_count[ni] = segn;
            }
            parent = ni;
            if (left) { hi = q; ni = li; saA = surface_area(AL); }
            else { lo = q; ni = ri; saA = surface_area(AR); }
#pragma unroll
            for (int k = 0; k < 6; k++) Cb[k] = Cn[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------------
template <typename T> void build_tree(bvhgpu_tree* t, const T* aabbs_dev, size_t n) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    bvhgpu_ctx* ctx = t->ctx;
    hipStream_t st = ctx->stream;
    t->built = false; t->flattened = false;
    t->n = n; t->n_nodes = n ? 2 * n - 1 : 0;
    t->n_flat = n >= 2 ? 3 * n - 2 : n;
    t->n_trav = n >= 2 ? 2 * n - 2 : n;
    t->levels = 0;
    if (n == 0) { t->built = true; return; }

    const size_t max_big = n / (SMALL_MAX + 1) + 2;
    const size_t max_tiles = n / TILE + max_big + 2;
    t->aabbs.reserve(n * 6 * sizeof(T));
    t->nodes.reserve(t->n_nodes * sizeof(typename Tr::Node));
    t->node_start.reserve(t->n_nodes * 4);
    t->node_count.reserve(t->n_nodes * 4);
    t->shape_node.reserve(n * 4);
    t->idx[0].reserve(n * 4);
    t->idx[1].reserve(n * 4);
    t->bk.reserve(n);
    for (int i = 0; i < 2; i++) {
        t->big[i].reserve(max_big * sizeof(Item<T>));
        t->stats[i].reserve(max_big * sizeof(ItemStats<T>));
        t->tile_item[i].reserve(max_tiles * 4);
    }
    t->small.reserve((n + 1) * sizeof(Item<T>));
    t->tile_cnt.reserve(max_tiles * NUM_BUCKETS * 4);
    t->ctr.reserve(ROOTKEY_OFF + STAT_KEYS * sizeof(Key));

    if ((const void*)aabbs_dev != t->aabbs.p)
        BVH_HIP(hipMemcpyAsync(t->aabbs.p, aabbs_dev, n * 6 * sizeof(T), hipMemcpyDeviceToDevice, st));

    BuildArgs<T> a;
    a.aabbs = t->aabbs.as<T>();
    a.nodes = t->nodes.as<typename Tr::Node>();
    a.node_start = t->node_start.as<uint32_t>();
    a.node_count = t->node_count.as<uint32_t>();
    a.shape_node = t->shape_node.as<uint32_t>();
    a.idx[0] = t->idx[0].as<uint32_t>(); a.idx[1] = t->idx[1].as<uint32_t>();
    a.bk = t->bk.as<uint8_t>();
    a.big[0] = t->big[0].as<Item<T>>(); a.big[1] = t->big[1].as<Item<T>>();
    a.small = t->small.as<Item<T>>();
    a.stats[0] = t->stats[0].as<ItemStats<T>>(); a.stats[1] = t->stats[1].as<ItemStats<T>>();
    a.tile_item[0] = t->tile_item[0].as<uint32_t>(); a.tile_item[1] = t->tile_item[1].as<uint32_t>();
    a.tile_cnt = t->tile_cnt.as<uint32_t>();
    a.ctr = t->ctr.as<uint32_t>();
    a.rootkeys = reinterpret_cast<Key*>(reinterpret_cast<char*>(t->ctr.p) + ROOTKEY_OFF);
    a.n = (uint32_t)n;

    hipLaunchKernelGGL(k_init<T>, dim3(1), dim3(256), 0, st, a);

    const int prep_grid = (int)std::min<size_t>((n + 255) / 256, (size_t)ctx->n_cu * 4);
    hipLaunchKernelGGL(k_prep<T>, dim3(prep_grid), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_root<T>, dim3(1), dim3(64), 0, st, a);

    const int tile_grid = (int)std::min<size_t>(max_tiles, 2048);
    const int sel_grid = (int)std::min<size_t>((max_big + 3) / 4, 1024);
    uint32_t* pin = reinterpret_cast<uint32_t*>(ctx->pinned);
    int level = 0;
    uint32_t n_small = 0;
    if (n > (size_t)SMALL_MAX) {
        // optimistic batch of levels without host round-trips, then one check; unbalanced trees
        // continue level by level
        int fixed = 3;
        for (size_t m = n; m > (size_t)SMALL_MAX; m = (m + 1) / 2) fixed++;
        if (fixed > MAXLV - 4) fixed = MAXLV - 4;
        auto run_level = [&](int L) {
            hipLaunchKernelGGL(k_bin<T>, dim3(tile_grid), dim3(256), 0, st, a, L);
            hipLaunchKernelGGL(k_select<T>, dim3(sel_grid), dim3(256), 0, st, a, L);
            hipLaunchKernelGGL(k_scatter<T>, dim3(tile_grid), dim3(256), 0, st, a, L);
        };
        for (; level < fixed; level++) run_level(level);
        while (true) {
            BVH_HIP(hipMemcpyAsync(pin, a.ctr, ROOTKEY_OFF, hipMemcpyDeviceToHost, st));
            BVH_HIP(hipStreamSynchronize(st));
            n_small = pin[CTR_SMALL];
            uint32_t pending = pin[CTR_LEVEL0 + 2 * lvl_slot(level)];
            if (pending == 0) break;
            if (level + 1 >= MAXLV - 2)  // recycle the slot the next level will append to
                BVH_HIP(hipMemsetAsync(a.ctr + CTR_LEVEL0 + 2 * lvl_slot(level + 1), 0, 8, st));
            run_level(level);
            level++;
        }
        // trailing empty levels of the optimistic batch do not count
        while (level > 0 && level - 1 < MAXLV - 2 && pin[CTR_LEVEL0 + 2 * (level - 1)] == 0) level--;
    } else {
        n_small = 1;
    }
    t->levels = level;
    if (n_small) hipLaunchKernelGGL(k_small<T>, dim3((n_small + 3) / 4), dim3(256), 0, st, a, n_small);
    BVH_HIP(hipGetLastError());
    t->built = true;
}

template void build_tree<float>(bvhgpu_tree*, const float*, size_t);
template void build_tree<double>(bvhgpu_tree*, const double*, size_t);

}  // namespace bvhgpu
