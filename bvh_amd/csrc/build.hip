// build.hip — top-down 6-bucket SAH builder of the reference (src/bvh/bvh_node.rs:81-279,
// src/bvh/bvh_impl.rs:53-96, src/utils.rs:59-109), re-designed for MI355X and bit-exact with it.
//
// The reference recurses node by node (rayon::join).  Node placement is arithmetic (pre-order:
// left = ni+1, right = ni+1+(2*nl-1), bvh_node.rs:138-142), so any node can be split as soon as
// its index slice is final, in any order.  Three tiers:
//
//  level tier     — nodes with more than 768 (small scenes) / 1536 shapes, level-synchronous: per level two launches over
//                   a queue of work items (one item = one BvhNodeBuildArgs).  A launch boundary is the cheapest
//                   grid-wide synchronisation on this chip (tools/ubench/gridbar.hip).
//                     k_bin   : one workgroup per 512-position tile of an item: bucket id per shape
//                               (bvh_node.rs:204-222), 6 x (count, AABB, centroid-AABB) reduced with LDS
//                               integer-key atomics, merged into one of the item's STAT_REP statistic replicas
//                               with global atomics (min/max are exact → order-free); per-tile bucket counts;
//                     k_split : two roles in one launch, because they are independent —
//                               select  : one wave per item: replicas merged, 5 candidate splits, strict-<
//                                         first-wins argmin (:231-247), writes the BvhNode, queues the children;
//                               scatter : stable bucket-major rewrite of the index slice (:250-272) = one stable
//                                         3-bit counting-sort pass (needs only the per-tile bucket counts, not
//                                         the chosen split): wave ballot ranks + counts of the earlier tiles.
//  workgroup tier — k_mid: a node of 65..768 (1536) shapes is finished by ONE workgroup with its index slice and AABBs
//                   in LDS, level by level, down to <= 64-shape children.
//  wave tier      — k_small: every node with <= 64 shapes is finished by ONE wavefront, one shape per lane, all
//                   levels of the subtree at once: segmented ballot ranks for the stable sort, ds_permute to move
//                   shapes, segmented min/max prefix+suffix scans for the L/R bounds of the 5 candidate splits
//                   (plays the role of rayon_executor's sequential cut-off, bvh_impl.rs:534).
#include "flatten_node.hpp"

namespace bvhgpu {

constexpr int MAXLV = 96;        // counter slots (levels beyond reuse the last two, host-synchronised)
constexpr int CTR_SMALL = 0;     // u32: number of small items (<= 64 shapes, wave-subtree tier)
// (u32 slot 3 is unused: it was the arrival ticket of a k_prep that created the root item itself)
constexpr int CTR_MID2 = 2;      // u32: number of workgroup-tier items (65 .. BuildArgs::mid_max shapes)
constexpr int CTR_FLAGS = 4;     // u32: BUILD_FLAG_* bits raised by the kernels, read back by the host with the counters
constexpr int CTR_TOPMASK = BUILD_CTR_TOPMASK;   // u32: bit h set when the level tier has written the BvhNode of heap number h < 16 (levels 0..3):
                                 // once bits 1..15 are there, the walk's item filter can run beside the rest of the build (traverse.hip k_wide_items)
constexpr uint32_t BUILD_FLAG_NONFINITE = 1u;    // a shape AABB holds NaN / ±inf, or the root centroid extent overflows: the
                                                 // reference panics there (bvh_node.rs:214-217, `to_usize().unwrap()`); nothing is built
constexpr uint32_t BUILD_FLAG_PERSIST_GAVE_UP = BSTAT_UNFINISHED;   // k_level<T, false, true>: a group barrier timed out (its workgroups were not all resident in time) or the
                                                 // subtree is deeper than the counter slots: the tree is unfinished, the host builds it again level by level
constexpr uint32_t BUILD_FLAG_EMPTY_SPLIT = 2u;  // some node had no winning SAH candidate (NaN / inf costs): its children carry
                                                 // Aabb::empty() bounds (bvh_node.rs:225-230), so a child box is NOT the join of its
                                                 // grandchildren and traversal must test every ancestor (no wide walk)
// Workgroup tier: nodes of 65 .. mid_max shapes, one workgroup per node (its AABBs live in 30-60 KB of
// LDS, so every CU runs several): it splits down to <= 64-shape children for the wave tier.  (A second, larger
// workgroup tier for 1025..4096 shapes existed until the level-synchronous tier got down to ~12 µs per level — below
// the ~20 µs per level a 4096-shape workgroup needs; see DESIGN.md.)
// Capacity / workgroup size of that tier, swept on create_n_cubes scenes (build ms, f32):
//   120 k triangles: 1024/256 0.245, 1024/512 0.239, 768/256 0.237, 768/384 0.232, 512/256 0.238, 1536/256 0.24, 2048/256 0.272
//   360 k: 768/384 0.424, 1024/256 0.368, 1536/256 0.383      1.2 M: 1.105 / 0.984 / 0.948      12 M: 10.9 / 9.56 / 9.07
// At 120 k there is about one such node per CU and the tier is a latency chain (more threads per node help); from a few hundred
// thousand triangles on there are several per CU and it is throughput (fewer, fatter workgroups and one level-tier pass less help).
// Hence two instantiations, chosen by the shape count at launch (BuildArgs::mid_max is the level tier's hand-over size).
#ifndef BVH_MID_SMALL_MAXN
#define BVH_MID_SMALL_MAXN 768
#endif
#ifndef BVH_MID_SMALL_THREADS
#define BVH_MID_SMALL_THREADS 384
#endif
template <typename T> struct MidSmallScene {   // up to MID_SCENE_SPLIT shapes
    static constexpr int MAXN = BVH_MID_SMALL_MAXN;
    static constexpr int THREADS = BVH_MID_SMALL_THREADS;
    static constexpr int HANDOFF = SMALL_MAX;
};
#ifndef BVH_MID_LARGE_MAXN
#define BVH_MID_LARGE_MAXN 1536
#endif
#ifndef BVH_MID_LARGE_THREADS
#define BVH_MID_LARGE_THREADS 256
#endif
template <typename T> struct MidLargeScene {
    static constexpr int MAXN = sizeof(T) == 4 ? BVH_MID_LARGE_MAXN : BVH_MID_LARGE_MAXN * 2 / 3;   // f64: 6 x 8 B per shape in LDS
    static constexpr int THREADS = BVH_MID_LARGE_THREADS;
    static constexpr int HANDOFF = SMALL_MAX;
};
constexpr size_t MID_SCENE_SPLIT = 250000;
#ifndef BVH_STAT_REP
#define BVH_STAT_REP 8
#endif
constexpr int STAT_REP = BVH_STAT_REP;   // global replicas of an item's statistics: tile t adds to replica t % 8, so the ~470 tiles of
                              // the root do not serialise on 78 addresses (k_bin of level 0: 13.4 -> see profiles); the
                              // selection merges the replicas
constexpr int CTR_LEVEL0 = 16;   // u32 pairs (n_items, n_tiles) per level slot
constexpr int CTR_XDIR = CTR_LEVEL0 + 2 * MAXLV;   // 8 x {kind, slot, start, count}: the tree's level-3 nodes (heap numbers 8 .. 15) as the level pass that
                                                   // created them left them: the subtree every workgroup group of k_level<T, false, true> owns
static_assert((CTR_XDIR + 8 * 4) * 4 <= 1024, "the directory lives inside the counter page");
constexpr size_t ROOTKEY_OFF = 1024;  // byte offset of k_prep's per-workgroup partial bounds (12 keys each) inside the ctr buffer
constexpr int PREP_MAX_WG = 1024;     // k_prep's grid never exceeds this

__host__ __device__ inline int lvl_slot(int L) { return L < MAXLV - 2 ? L : (MAXLV - 2 + (L & 1)); }

template <typename T> struct ItemStats {
    typename Traits<T>::Key k[NUM_BUCKETS * STAT_KEYS];
    uint32_t cnt[NUM_BUCKETS];
    uint32_t _pad[2];
};

// Addressing of the level tier when a level is ONE launch (k_level): nothing a workgroup needs during the launch may come out
// of an atomic another workgroup of the same launch executes, so queue slots and tile numbers are ARITHMETIC:
//   item slot  = start / slot_div          slot_div = mid_max + 1: two items of a level are disjoint slices of more than
//                                          mid_max positions each, so their slots differ
//   tile id    = start / TILE + slot + k   k-th tile of the item; strictly increasing over a level's items (the slot term
//                                          separates the last tile of an item from the first of the next)
// and the arrays a level accumulates into (statistics, per-tile bucket counts, tile → item map) rotate over three buffers:
// level L reads the parents' [(L-1) % 3], accumulates into [L % 3] and resets [(L+1) % 3] for the launch after it.
template <typename T> struct LevelArgs {
    Item<T>* item[2];        // [L & 1][slot]
    uint4* tile_map[3];      // [L % 3][tile id] = {item slot (NONE: no such tile at this level), start, count, 0}
    ItemStats<T>* stats[3];  // [L % 3][slot * STAT_REP + replica]
    uint32_t* tile_cnt[3];   // [L % 3][tile id * NUM_BUCKETS + bucket]
    uint8_t* bk[2];          // [L & 1][position]: bucket of the shape at that position
    uint32_t slot_div, n_slots, n_tiles;
};

template <typename T> struct BuildArgs {
    LevelArgs<T> lv;
    uint32_t mid_max;        // nodes with at most this many shapes (and more than 64) go to the workgroup tier
    const T* aabbs;          // the tree's own copy (what every later kernel gathers from)
    const T* src;            // the caller's array: k_prep copies it into `aabbs` while it reduces the bounds
    typename Traits<T>::Node* nodes;
    uint32_t* node_start;
    uint32_t* node_count;
    uint32_t* shape_node;
    uint16_t* node_slot;     // heap number of every node (SLOT_NONE beyond the first 15 levels): traversal's LDS slots
    uint32_t* slot_entry;    // cleared here, filled by flatten
    uint32_t n_slots;
    uint32_t* wslot_node;    // WIDE_SLOTS entries: LDS slot table of the wide walk, cleared here, filled by flatten
    uint32_t* idx[2];
    uint8_t* bk;
    Item<T>* big[2];
    Item<T>* mid2;
    Item<T>* small;
    ItemStats<T>* stats[2];
    uint32_t* tile_item[2];
    uint32_t* tile_cnt;
    uint32_t* chunk_cnt[2];  // two-launch schedule, items of more than CHUNK_TILES tiles: bucket counts per CHUNK_TILES-tile block of the level's tile
                             // ids, [level & 1][(block * 2 + slot) * NUM_BUCKETS + bucket] (slot 1: the item starts inside the block) — see scatter_role
    uint32_t n_chunks;       // blocks per parity
    FlattenArgs<T> fl;       // k_small, fl_parts != 0: the wave that has built a subtree also writes these parts of the flatten for its nodes (flatten_node.hpp)
    uint32_t fl_parts;
    uint32_t tile2;          // positions per tile in the two-launch schedule (level_tile(): TILE, more on scenes of millions of shapes); the fused schedule keeps TILE
    uint32_t* ctr;
    typename Traits<T>::Key* rootkeys;   // [gridDim of k_prep][12]: every workgroup's bounds (joined by k_root / k_level<ROOT>)
    uint32_t prep_wgs;                   // gridDim of k_prep
    uint32_t n;
    unsigned long long* xbar;            // k_level<T, false, true>: per workgroup group two 128-byte lines {arrivals | live << 32} and {round | live << 32}; zeroed by k_prep
};
constexpr int XBAR_WORDS = 8 * 32;       // unsigned long long words: 8 groups x 2 lines of 16

// is key slot j of a bucket a "min" slot?  layout: aabb.min[3] aabb.max[3] cen.min[3] cen.max[3]
__device__ __forceinline__ bool key_is_min(int j) { return j < 3 || (j >= 6 && j < 9); }

// ------------------------------------------------------------------------------------------------
// K1 prep: identity permutation (bvh_impl.rs:61-63) + joint_aabb_of_shapes over all shapes (:74)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ void push_item(const BuildArgs<T>& a, int next_level, uint32_t ni, uint32_t parent, uint32_t start,
                          uint32_t count, const T* A, const T* C, uint32_t heap, int lane);
template <typename T> __device__ void init_stats(ItemStats<T>* s, int lane);

template <typename T> __global__ __launch_bounds__(256) void k_prep(BuildArgs<T> a) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    __shared__ Key sk[STAT_KEYS];
    if (threadIdx.x < STAT_KEYS) sk[threadIdx.x] = key_is_min(threadIdx.x) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
    __syncthreads();
    T loc[STAT_KEYS];   // joined on floats (one v_min/v_max each); keys only for the atomics that merge waves
#pragma unroll
    for (int j = 0; j < STAT_KEYS; j++) loc[j] = key_is_min(j) ? Tr::inf() : -Tr::inf();
    if (a.lv.tile_map[0]) {   // level tier, one launch per level: tile maps and tile counts of all three rotating buffers
        const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, gs = gridDim.x * blockDim.x;
        for (uint32_t i = gt; i < 3u * a.lv.n_tiles; i += gs) a.lv.tile_map[i / a.lv.n_tiles][i % a.lv.n_tiles] = make_uint4(NONE, 0u, 0u, 0u);
        const uint32_t nc = a.lv.n_tiles * (uint32_t)NUM_BUCKETS;
        for (uint32_t i = gt; i < 3u * nc; i += gs) a.lv.tile_cnt[i / nc][i % nc] = 0u;
    }
    if (a.lv.tile_map[0] && blockIdx.x == 0 && threadIdx.x < WAVE)   // ... and the root's statistic replicas (k_level<ROOT> adds to them)
        for (int r = 0; r < STAT_REP; r++) init_stats<T>(&a.lv.stats[0][r], (int)threadIdx.x);
    if (a.xbar && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < (uint32_t)XBAR_WORDS; i += blockDim.x) a.xbar[i] = 0ull;
    if (a.chunk_cnt[0] && blockIdx.x == 1 % gridDim.x)   // both parities start a build all-zero (k_split keeps the next level's so)
        for (uint32_t i = threadIdx.x; i < 2u * a.n_chunks * 2u * (uint32_t)NUM_BUCKETS; i += blockDim.x) a.chunk_cnt[0][i] = 0u;
    if (blockIdx.x == 0) {   // the LDS slot tables of the previous tree (filled again by flatten)
        for (uint32_t i = threadIdx.x; i < a.n_slots; i += blockDim.x) a.slot_entry[i] = NONE;
        for (uint32_t i = threadIdx.x; i < WIDE_SLOTS; i += blockDim.x) a.wslot_node[i] = NONE;
    }
    const bool copy = a.src != a.aabbs;
    T* own = const_cast<T*>(a.aabbs);
    bool bad = false;   // input contract: the reference panics on NaN / inf centroids (bvh_node.rs:214-217); a single shape is never bucketed
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
        a.idx[0][i] = i;
        const T* b = a.src + 6 * (size_t)i;
        T bx[6];
#pragma unroll
        for (int k = 0; k < 6; k++) bx[k] = b[k];
#pragma unroll
        for (int k = 0; k < 6; k++) bad = bad || !(fabs(bx[k]) < Tr::inf());   // NaN or ±inf (the comparison is false for NaN)
        if (copy) {
#pragma unroll
            for (int k = 0; k < 6; k++) own[6 * (size_t)i + k] = bx[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const T c = center1(bx[k], bx[3 + k]);
            loc[k] = join_min(loc[k], bx[k]);
            loc[3 + k] = join_max(loc[3 + k], bx[3 + k]);
            loc[6 + k] = join_min(loc[6 + k], c);
            loc[9 + k] = join_max(loc[9 + k], c);
        }
    }
    {   // wave butterfly (all 12 moves of a step in one batch), then one LDS key atomic per wave and value
        const int lane = lane_id();
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            T u[STAT_KEYS];
#pragma unroll
            for (int j = 0; j < STAT_KEYS; j++) u[j] = lane_fetch(loc[j], (lane ^ d) << 2);
#pragma unroll
            for (int j = 0; j < STAT_KEYS; j++) loc[j] = key_is_min(j) ? join_min(loc[j], u[j]) : join_max(loc[j], u[j]);
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < STAT_KEYS; j++) {
                if (key_is_min(j)) atomicMin(&sk[j], Tr::key(loc[j]));
                else atomicMax(&sk[j], Tr::key(loc[j]));
            }
        }
    }
    if (a.n > 1 && __any(bad) && lane_id() == 0) atomicOr(&a.ctr[CTR_FLAGS], BUILD_FLAG_NONFINITE);
    __syncthreads();
    // Every workgroup leaves its 12 bounds in its own row; k_root joins the rows.  (History: 12 global min / max atomics per
    // workgroup on one cache line; then plain rows + fence + ticket with the last workgroup creating the root item in this
    // launch — measured with tools/prep_diag.sh: this kernel 7 µs, the agent-scope fence +6.5 µs, ticket + last-workgroup
    // phase +7 µs.  A dependent launch of one workgroup costs less than either.)
    if (threadIdx.x < STAT_KEYS) a.rootkeys[(size_t)blockIdx.x * STAT_KEYS + threadIdx.x] = sk[threadIdx.x];
}

// joint_aabb_of_shapes finished (bvh_impl.rs:74) → the root work item (BvhNodeBuildArgs, bvh_impl.rs:75-87).  One workgroup.
template <typename T> __global__ __launch_bounds__(256) void k_root(BuildArgs<T> a, uint32_t prep_wgs, int fused) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    __shared__ Key sk[STAT_KEYS];
    if (threadIdx.x < STAT_KEYS) sk[threadIdx.x] = key_is_min(threadIdx.x) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < prep_wgs * (uint32_t)STAT_KEYS; e += blockDim.x) {   // rows x keys, coalesced
        const Key v = a.rootkeys[e];
        const int j = (int)(e % (uint32_t)STAT_KEYS);
        if (key_is_min(j)) atomicMin(&sk[j], v);
        else atomicMax(&sk[j], v);
    }
    __syncthreads();
    if (threadIdx.x < WAVE) {
        T A[6], C[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { A[k] = Tr::unkey(sk[k]); C[k] = Tr::unkey(sk[6 + k]); }
        uint32_t flags = a.ctr[CTR_FLAGS];
        if (a.n > 1) {   // finite boxes whose centroid extent overflows: (c - cmin) / ext is NaN for some shape → same panic
            bool ovf = false;
#pragma unroll
            for (int k = 0; k < 3; k++) ovf = ovf || !(fabs(C[3 + k] - C[k]) < Tr::inf());
            if (ovf) { flags |= BUILD_FLAG_NONFINITE; if (threadIdx.x == 0) atomicOr(&a.ctr[CTR_FLAGS], BUILD_FLAG_NONFINITE); }
        }
        // invalid input: no root item, so every later kernel of the optimistic schedule finds empty queues
        if (flags & BUILD_FLAG_NONFINITE) return;
        if (fused && a.n > a.mid_max) {   // level tier, one launch per level: the root is item 0 of level 0, its tiles are 0 .. ntile-1
            const int lane = (int)threadIdx.x;
            Item<T>* it = &a.lv.item[0][0];
            if (lane == 0) {
                it->ni = 0; it->parent = 0; it->start = 0; it->count = a.n; it->tile_base = 0; it->parity = 0; it->heap = 1u; it->_r1 = 0;
                a.ctr[CTR_LEVEL0] = 1u;
            }
            if (lane < 6) { it->A[lane] = A[lane]; it->C[lane] = C[lane]; }
            const uint32_t ntile = (a.n + TILE - 1) / TILE;
            for (uint32_t j = lane; j < ntile; j += WAVE) a.lv.tile_map[0][j] = make_uint4(0u, 0u, a.n, 0u);
            for (int r = 0; r < STAT_REP; r++) init_stats<T>(&a.lv.stats[0][r], lane);
        } else {
            push_item<T>(a, 0, 0u, 0u, 0u, a.n, A, C, 1u, (int)threadIdx.x);
        }
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
                          uint32_t count, const T* A, const T* C, uint32_t heap, int lane) {
    const int nslot = lvl_slot(next_level);
    const int npar = next_level & 1;
    uint32_t slot = 0, tb = 0;
    const bool is_small = count <= (uint32_t)SMALL_MAX;
    const bool is_mid2 = !is_small && count <= a.mid_max;
    const uint32_t ntile = (count + a.tile2 - 1) / a.tile2;
    if (lane == 0) {
        if (is_small) slot = atomicAdd(&a.ctr[CTR_SMALL], 1u);
        else if (is_mid2) slot = atomicAdd(&a.ctr[CTR_MID2], 1u);
        else {
            slot = atomicAdd(&a.ctr[CTR_LEVEL0 + 2 * nslot], 1u);
            tb = atomicAdd(&a.ctr[CTR_LEVEL0 + 2 * nslot + 1], ntile);
        }
    }
    slot = __shfl(slot, 0);
    tb = __shfl(tb, 0);
    Item<T>* it = is_small ? &a.small[slot] : (is_mid2 ? &a.mid2[slot] : &a.big[npar][slot]);
    if (lane == 0) {
        it->ni = ni; it->parent = parent; it->start = start; it->count = count;
        it->tile_base = tb; it->parity = (uint32_t)npar; it->heap = heap; it->_r1 = 0;
    }
    if (lane < 6) { it->A[lane] = A[lane]; it->C[lane] = C[lane]; }
    if (!is_small && !is_mid2) {
        for (uint32_t j = lane; j < ntile; j += WAVE) a.tile_item[npar][tb + j] = slot;
        for (int r = 0; r < STAT_REP; r++) init_stats<T>(&a.stats[npar][(size_t)slot * STAT_REP + r], lane);
    }
}

// enqueue BOTH children of a node.  The queue slots are reserved by lanes 0 and 1 at the same time (one atomic
// each; a level-tier child takes its item slot and its tile range with ONE 64-bit add on the adjacent
// (items, tiles) counters), so the select kernel pays one atomic round trip per node instead of four.
template <typename T>
__device__ void push_pair(const BuildArgs<T>& a, int next_level, uint32_t parent, uint32_t li, uint32_t lstart, uint32_t lcount,
                          const T* AL, const T* CL, uint32_t lheap, uint32_t ri, uint32_t rstart, uint32_t rcount,
                          const T* AR, const T* CR, uint32_t rheap, int lane) {
    const int nslot = lvl_slot(next_level);
    const int npar = next_level & 1;
    const uint32_t mycount = lane == 0 ? lcount : rcount;
    const int mykind = mycount <= (uint32_t)SMALL_MAX ? 0 : (mycount <= a.mid_max ? 1 : 3);   // wave / workgroup / level tier
    const uint32_t myntile = (mycount + a.tile2 - 1) / a.tile2;
    uint32_t slot = 0, tb = 0;
    if (lane < 2) {
        if (mykind == 0) slot = atomicAdd(&a.ctr[CTR_SMALL], 1u);
        else if (mykind == 1) slot = atomicAdd(&a.ctr[CTR_MID2], 1u);
        else {
            const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(&a.ctr[CTR_LEVEL0 + 2 * nslot]),
                                                     1ull | ((unsigned long long)myntile << 32));
            slot = (uint32_t)old; tb = (uint32_t)(old >> 32);
        }
    }
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const uint32_t cslot = __shfl(slot, side), ctb = __shfl(tb, side);
        const int kind = __shfl(mykind, side);
        const uint32_t ccount = side ? rcount : lcount, ntile = (ccount + a.tile2 - 1) / a.tile2;
        Item<T>* it = kind == 0 ? &a.small[cslot] : (kind == 1 ? &a.mid2[cslot] : &a.big[npar][cslot]);
        if (lane == 0) {
            it->ni = side ? ri : li; it->parent = parent; it->start = side ? rstart : lstart; it->count = ccount;
            it->tile_base = ctb; it->parity = (uint32_t)npar; it->heap = side ? rheap : lheap; it->_r1 = 0;
        }
        if (lane < 6) { it->A[lane] = side ? AR[lane] : AL[lane]; it->C[lane] = side ? CR[lane] : CL[lane]; }
        if (kind == 3) {
            for (uint32_t j = lane; j < ntile; j += WAVE) a.tile_item[npar][ctb + j] = cslot;
            for (int r = 0; r < STAT_REP; r++) init_stats<T>(&a.stats[npar][(size_t)cslot * STAT_REP + r], lane);
        }
    }
}

// counters = 0
template <typename T> __global__ __launch_bounds__(256) void k_init(BuildArgs<T> a) {
    for (int i = threadIdx.x; i < (int)(ROOTKEY_OFF / 4); i += 256) a.ctr[i] = 0;
}

// The build's counters go to the tree's pinned host page and are reset for the next build in the same launch (the
// runtime's copy kernel plus k_init cost ~4.5 µs each on the stream; this is one ~4 µs launch)
template <typename T> __global__ __launch_bounds__(256) void k_publish_build(BuildArgs<T> a, uint32_t* __restrict__ host_page) {
    for (int i = threadIdx.x; i < (int)(ROOTKEY_OFF / 4); i += 256) { host_page[i] = a.ctr[i]; a.ctr[i] = 0; }
    __threadfence_system();
}

// ------------------------------------------------------------------------------------------------
// tier 1 / bin
// ------------------------------------------------------------------------------------------------
#ifndef BVH_CHUNK_TILES
#define BVH_CHUNK_TILES 256
#endif
constexpr int SCATTER_AHEAD = 4;   // rounds of 256 positions whose loads the stable scatter issues together
constexpr int CHUNK_TILES = BVH_CHUNK_TILES;   // tile ids per block of BuildArgs::chunk_cnt (items above CHUNK_TILES tiles use the block sums)
constexpr int BIN_REP = 16;  // LDS replicas of the tile statistics: lanes l and l+16.. share one, so a wave's
                             // same-address atomic conflicts drop from ~64/6 to ~4/6 per instruction
template <typename T> __global__ __launch_bounds__(256) void k_bin(BuildArgs<T> a, int level) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    const int slot = lvl_slot(level), par = level & 1;
    const uint32_t ntiles = a.ctr[CTR_LEVEL0 + 2 * slot + 1];
    __shared__ Key sk[BIN_REP][NUM_BUCKETS * STAT_KEYS];
    __shared__ uint32_t sc[BIN_REP][NUM_BUCKETS];
    const uint32_t* idx = a.idx[par];
    const int rep = threadIdx.x & (BIN_REP - 1);
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t item_id = a.tile_item[par][t];
        const Item<T>* it = &a.big[par][item_id];
        const uint32_t start = it->start, count = it->count;
        const uint32_t p0 = start + (t - it->tile_base) * a.tile2;
        const uint32_t pend = min(start + count, p0 + a.tile2);
        T C[6];
#pragma unroll
        for (int k = 0; k < 6; k++) C[k] = it->C[k];
        const int ax = largest_axis(C);                 // bvh_node.rs:107
        const T cmin = C[ax];
        const T ext = C[3 + ax] - C[ax];                // :108
        const bool degen = ext < Tr::eps();             // :114
        const uint32_t half = count / 2;                // :117
        for (int j = threadIdx.x; j < BIN_REP * NUM_BUCKETS * STAT_KEYS; j += 256)
            (&sk[0][0])[j] = key_is_min(j % STAT_KEYS) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
        if (threadIdx.x < BIN_REP * NUM_BUCKETS) (&sc[0][0])[threadIdx.x] = 0;
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
            Key* kk = &sk[rep][bkt * STAT_KEYS];           // Bucket::add_aabb (utils.rs:81-85)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                atomicMin(&kk[k], Tr::key(bx[k]));
                atomicMax(&kk[3 + k], Tr::key(bx[3 + k]));
                Key kc = Tr::key(c[k]);
                atomicMin(&kk[6 + k], kc);
                atomicMax(&kk[9 + k], kc);
            }
            atomicAdd(&sc[rep][bkt], 1u);
        }
        __syncthreads();
        ItemStats<T>* gs = &a.stats[par][(size_t)item_id * STAT_REP + ((t - it->tile_base) & (STAT_REP - 1))];
        if (threadIdx.x < NUM_BUCKETS) {
            uint32_t c = 0;
#pragma unroll
            for (int r = 0; r < BIN_REP; r++) c += sc[r][threadIdx.x];
            sc[0][threadIdx.x] = c;   // only this thread reads/writes column threadIdx.x here
            a.tile_cnt[t * NUM_BUCKETS + threadIdx.x] = c;
            if (c) atomicAdd(&gs->cnt[threadIdx.x], c);
            if (c && count > (uint32_t)CHUNK_TILES * a.tile2) {   // an item of many tiles: its scatter adds up block sums instead of every earlier tile's counts
                const uint32_t blk = t / (uint32_t)CHUNK_TILES;
                const uint32_t slot = it->tile_base > blk * (uint32_t)CHUNK_TILES ? 1u : 0u;
                atomicAdd(&a.chunk_cnt[par][(size_t)(blk * 2u + slot) * NUM_BUCKETS + threadIdx.x], c);
            }
        }
        __syncthreads();
        if (threadIdx.x < NUM_BUCKETS * STAT_KEYS) {
            const int j = threadIdx.x;
            if (sc[0][j / STAT_KEYS]) {
                Key v = sk[0][j];
                const bool mn = key_is_min(j % STAT_KEYS);
#pragma unroll
                for (int r = 1; r < BIN_REP; r++) { const Key u = sk[r][j]; v = mn ? (u < v ? u : v) : (u > v ? u : v); }
                if (mn) atomicMin(&gs->k[j], v);
                else atomicMax(&gs->k[j], v);
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

// ------------------------------------------------------------------------------------------------
// SAH split selection from the 6 bucket statistics (bvh_node.rs:224-247).  keys: 6 x 12 integer keys
// (aabb min3 max3, centroid min3 max3), cnts: 6 counts.  In the degenerate branch (:114-124) "bucket"
// 0 / 1 are the two halves of the index list and the split is forced between them, which reproduces
// joint_aabb_of_shapes of each half (:118-119).  Returns n_left; fills child bounds and cnt[].
// ------------------------------------------------------------------------------------------------
template <typename T, typename KeyPtr, typename CntPtr>
__device__ __forceinline__ uint32_t sah_select(KeyPtr keys, CntPtr cnts, const T* A, bool degen, uint32_t* cnt, T* AL,
                                               T* CL, T* AR, T* CR, bool& no_winner) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    // Joins are exact, so fold(empty, join) over buckets 0..s / s+1..5 (utils.rs:88-94) equals running
    // prefix / suffix joins; they are done on the monotone integer keys (one v_min/v_max_u32 each, and
    // the -0 < +0 order of the float joins for free) and decoded only where a float is needed.
    Key pa[NUM_BUCKETS][6], sa[NUM_BUCKETS][6];   // aabb prefix (buckets 0..b) / suffix (b..5) joins
#pragma unroll
    for (int b = 0; b < NUM_BUCKETS; b++) {
        cnt[b] = cnts[b];
#pragma unroll
        for (int k = 0; k < 6; k++) { pa[b][k] = keys[b * STAT_KEYS + k]; sa[b][k] = pa[b][k]; }
    }
#pragma unroll
    for (int b = 1; b < NUM_BUCKETS; b++) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            pa[b][k] = pa[b][k] < pa[b - 1][k] ? pa[b][k] : pa[b - 1][k];
            pa[b][3 + k] = pa[b][3 + k] > pa[b - 1][3 + k] ? pa[b][3 + k] : pa[b - 1][3 + k];
        }
    }
#pragma unroll
    for (int b = NUM_BUCKETS - 2; b >= 0; b--) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            sa[b][k] = sa[b][k] < sa[b + 1][k] ? sa[b][k] : sa[b + 1][k];
            sa[b][3 + k] = sa[b][3 + k] > sa[b + 1][3 + k] ? sa[b][3 + k] : sa[b + 1][3 + k];
        }
    }
    T min_cost = Tr::inf();
    int best = -1;   // no winner (NaN/inf costs): the reference keeps min_bucket = 0 and EMPTY child bounds (:225-230)
    const T sa_parent = surface_area(A);
    uint32_t ln = 0, total = 0;
#pragma unroll
    for (int b = 0; b < NUM_BUCKETS; b++) total += cnt[b];
#pragma unroll
    for (int s = 0; s < NUM_BUCKETS - 1; s++) {
        ln += cnt[s];
        T la[6], ra[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { la[k] = Tr::unkey(pa[s][k]); ra[k] = Tr::unkey(sa[s + 1][k]); }
        T cl = (T)ln * surface_area(la);
        T cr = (T)(total - ln) * surface_area(ra);
        T num = cl + cr;
        T cost = num / sa_parent;                               // :236-238
        bool take = degen ? (s == 0) : (cost < min_cost);       // strict <, first wins (:239)
        if (take) { min_cost = cost; best = s; }
    }
    no_winner = best < 0;
    if (best < 0) {
        box_empty(AL); box_empty(CL); box_empty(AR); box_empty(CR);
        return cnt[0];
    }
    // child bounds of the winning split: aabb from the running joins, centroid bounds joined now
    Key cl_[6], cr_[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        cl_[k] = key_is_min(k) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
        cr_[k] = cl_[k];
    }
    uint32_t nl = 0;
#pragma unroll
    for (int b = 0; b < NUM_BUCKETS; b++) {
        const bool left = b <= best;
        if (left) nl += cnt[b];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const Key v = keys[b * STAT_KEYS + 6 + k];
            const Key jl = key_is_min(k) ? (v < cl_[k] ? v : cl_[k]) : (v > cl_[k] ? v : cl_[k]);
            const Key jr = key_is_min(k) ? (v < cr_[k] ? v : cr_[k]) : (v > cr_[k] ? v : cr_[k]);
            cl_[k] = left ? jl : cl_[k];
            cr_[k] = left ? cr_[k] : jr;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        // pa/sa are indexed by a run-time `best`: select with a compare chain, not a dynamic register index
        Key l = pa[0][k], r = sa[1][k];
#pragma unroll
        for (int s = 1; s < NUM_BUCKETS - 1; s++) { if (best == s) { l = pa[s][k]; r = sa[s + 1][k]; } }
        AL[k] = Tr::unkey(l); AR[k] = Tr::unkey(r);
        CL[k] = Tr::unkey(cl_[k]); CR[k] = Tr::unkey(cr_[k]);
    }
    return nl;
}

template <typename T> __device__ void select_role(const BuildArgs<T>& a, int level, uint32_t block, uint32_t nblocks) {
    using Tr = Traits<T>;
    const int slot = lvl_slot(level), par = level & 1;
    const uint32_t nitems = a.ctr[CTR_LEVEL0 + 2 * slot];
    const int lane = lane_id();
    const uint32_t wave0 = (block * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (nblocks * blockDim.x) >> 6;
    for (uint32_t id = wave0; id < nitems; id += nwaves) {
        const Item<T>* it = &a.big[par][id];
        // merge the STAT_REP replicas the tiles added to (joins are exact, counts are integers): lane j owns keys j, j+64
        __shared__ ItemStats<T> s_merged[4];
        ItemStats<T>* st = &s_merged[threadIdx.x >> 6];
        {
            using Key = typename Tr::Key;
            const ItemStats<T>* rep = &a.stats[par][(size_t)id * STAT_REP];
            for (int j = lane; j < NUM_BUCKETS * STAT_KEYS; j += WAVE) {
                const bool mn = key_is_min(j % STAT_KEYS);
                Key v = rep[0].k[j];
#pragma unroll
                for (int r = 1; r < STAT_REP; r++) { const Key u = rep[r].k[j]; v = mn ? (u < v ? u : v) : (u > v ? u : v); }
                st->k[j] = v;
            }
            if (lane < NUM_BUCKETS) {
                uint32_t c = 0;
#pragma unroll
                for (int r = 0; r < STAT_REP; r++) c += rep[r].cnt[lane];
                st->cnt[lane] = c;
            }
        }
        const uint32_t ni = it->ni, parent = it->parent, start = it->start, count = it->count;
        T A[6], C[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { A[k] = it->A[k]; C[k] = it->C[k]; }
        const int ax = largest_axis(C);
        const T ext = C[3 + ax] - C[ax];
        const bool degen = ext < Tr::eps();

        uint32_t cnt[NUM_BUCKETS];
        T AL[6], CL[6], AR[6], CR[6];
        bool no_winner;
        const uint32_t nl = sah_select<T>(st->k, st->cnt, A, degen, cnt, AL, CL, AR, CR, no_winner);
        if (no_winner && lane == 0) atomicOr(&a.ctr[CTR_FLAGS], BUILD_FLAG_EMPTY_SPLIT);
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
            a.node_slot[ni] = (uint16_t)it->heap;
            if (it->heap < 16u) atomicOr(&a.ctr[CTR_TOPMASK], 1u << it->heap);
        }
        push_pair<T>(a, level + 1, ni, li, start, nl, AL, CL, heap_child(it->heap, 0u), ri, start + nl, count - nl, AR, CR,
                     heap_child(it->heap, 1u), lane);

    }
}

// ------------------------------------------------------------------------------------------------
// tier 1 / scatter — stable bucket-major rewrite (bvh_node.rs:250-272)
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ void scatter_role(const BuildArgs<T>& a, int level, uint32_t block, uint32_t nblocks) {
    const int slot = lvl_slot(level), par = level & 1;
    const uint32_t ntiles = a.ctr[CTR_LEVEL0 + 2 * slot + 1];
    __shared__ uint32_t run[NUM_BUCKETS];
    __shared__ uint32_t wcnt[4][NUM_BUCKETS];
    __shared__ __attribute__((aligned(16))) uint32_t wcnt2[2][NUM_BUCKETS][4];   // [round & 1][bucket][wave]
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const unsigned long long lt = lanemask_lt();
    const uint32_t* src = a.idx[par];
    uint32_t* dst = a.idx[par ^ 1];
    for (uint32_t t = block; t < ntiles; t += nblocks) {
        const Item<T>* it = &a.big[par][a.tile_item[par][t]];
        const uint32_t start = it->start, count = it->count;
        const uint32_t p0 = start + (t - it->tile_base) * a.tile2;
        const uint32_t pend = min(start + count, p0 + a.tile2);
        // exclusive offset of (this tile, bucket b) inside the item's slice = shapes of the item in buckets < b
        // + shapes of bucket b in the item's earlier tiles.  Every workgroup adds those up itself (at most a few
        // hundred tiles per item) — a serial scan per item in the selection was the longest part of that launch.
        {
            const uint32_t tl = t - it->tile_base, ntl = (count + a.tile2 - 1) / a.tile2;
            uint32_t before[NUM_BUCKETS], all[NUM_BUCKETS];
#pragma unroll
            for (int b = 0; b < NUM_BUCKETS; b++) { before[b] = 0; all[b] = 0; }
            const uint32_t* tc = a.tile_cnt + (size_t)it->tile_base * NUM_BUCKETS;
            if (count > (uint32_t)CHUNK_TILES * a.tile2) {
                // An item of MANY tiles (the root of a 12 M-shape scene has 23 437): every one of its workgroups adding up all of its tiles' counts is
                // quadratic — 13 GB of (cached) reads for that root alone, k_split 0.72 ms at the top levels against 0.07 ms further down.  k_bin
                // has left the counts per block of CHUNK_TILES tile ids as well (a block holds tiles of at most two such items: the tail of one —
                // slot 0 — and the head of the next — slot 1): whole blocks by their sums, the ragged ends tile by tile.
                const uint32_t T0 = it->tile_base, T1 = T0 + ntl;
                const uint32_t g0 = T0 / (uint32_t)CHUNK_TILES, g1 = (T1 - 1u) / (uint32_t)CHUNK_TILES, gt = t / (uint32_t)CHUNK_TILES;
                const uint32_t* cc = a.chunk_cnt[par];
                for (uint32_t g = g0 + threadIdx.x; g <= g1; g += 256) {
                    const uint32_t sl = T0 > g * (uint32_t)CHUNK_TILES ? 1u : 0u;
#pragma unroll
                    for (int b = 0; b < NUM_BUCKETS; b++) {
                        const uint32_t v = cc[(size_t)(g * 2u + sl) * NUM_BUCKETS + b];
                        all[b] += v;
                        before[b] += g < gt ? v : 0u;
                    }
                }
                // the tiles of this tile's own block that come before it
                const uint32_t j0 = max(gt * (uint32_t)CHUNK_TILES, T0);
                for (uint32_t j = j0 + threadIdx.x; j < t; j += 256) {
#pragma unroll
                    for (int b = 0; b < NUM_BUCKETS; b++) before[b] += a.tile_cnt[(size_t)j * NUM_BUCKETS + b];
                }
            } else {
                for (uint32_t j = threadIdx.x; j < ntl; j += 256) {
#pragma unroll
                    for (int b = 0; b < NUM_BUCKETS; b++) {
                        const uint32_t v = tc[(size_t)j * NUM_BUCKETS + b];
                        all[b] += v;
                        before[b] += j < tl ? v : 0u;
                    }
                }
            }
            // run[b] = (shapes of the item in buckets < b) + (shapes of bucket b in earlier tiles)
            uint32_t mine[NUM_BUCKETS];
            uint32_t acc = 0;
#pragma unroll
            for (int b = 0; b < NUM_BUCKETS; b++) { mine[b] = acc + before[b]; acc += all[b]; }
            // (acc over `all` is a per-thread partial: the sum over threads of (acc_prefix + before) is what we need)
#pragma unroll
            for (int b = 0; b < NUM_BUCKETS; b++) {
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) mine[b] += __shfl_down(mine[b], d);
            }
            if (lane == 0) {
#pragma unroll
                for (int b = 0; b < NUM_BUCKETS; b++) wcnt[w][b] = mine[b];
            }
            __syncthreads();
            if (threadIdx.x < NUM_BUCKETS)
                run[threadIdx.x] = wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        }
        __syncthreads();
        // One barrier per 256 positions: the waves' bucket counts go to one of two LDS sets in turn (a wave is never more than one barrier ahead
        // of the slowest), and every thread keeps the running offsets of all six buckets itself — three barriers per round cost a 4096-position
        // tile 48 of them (k_split at 12 M shapes: 86 µs a level).
        uint32_t runl[NUM_BUCKETS];
#pragma unroll
        for (int bb = 0; bb < NUM_BUCKETS; bb++) runl[bb] = run[bb];
        // ... and the loads of SCATTER_AHEAD rounds are issued together: one 4-byte + one 1-byte load in flight per thread moved a level's 108 MB
        // at 1.25 TB/s (24 waves per CU x 320 B against ~2 µs of latency)
        uint32_t round = 0;
        for (uint32_t c0 = p0; c0 < pend; c0 += 256u * SCATTER_AHEAD) {
            int bq[SCATTER_AHEAD];
            uint32_t sq[SCATTER_AHEAD];
#pragma unroll
            for (int u = 0; u < SCATTER_AHEAD; u++) {
                const uint32_t p = c0 + 256u * (uint32_t)u + threadIdx.x;
                bq[u] = p < pend ? (int)a.bk[p] : 7;
                sq[u] = p < pend ? src[p] : 0u;
            }
#pragma unroll
            for (int u = 0; u < SCATTER_AHEAD; u++) {
                if (c0 + 256u * (uint32_t)u >= pend) break;   // (workgroup-uniform)
                const int b = bq[u];
                const bool valid = b != 7;
                uint32_t (*wc)[4] = wcnt2[round & 1u];
                round++;
                uint32_t rank = 0;
#pragma unroll
                for (int bb = 0; bb < NUM_BUCKETS; bb++) {
                    unsigned long long m = __ballot(b == bb);
                    if (b == bb) rank = (uint32_t)__popcll(m & lt);
                    if (lane == 0) wc[bb][w] = (uint32_t)__popcll(m);
                }
                __syncthreads();
                uint32_t mine = 0, before_w = 0;
#pragma unroll
                for (int bb = 0; bb < NUM_BUCKETS; bb++) {
                    const uint4 c = *reinterpret_cast<const uint4*>(&wc[bb][0]);
                    if (b == bb) { mine = runl[bb]; before_w = (w > 0 ? c.x : 0u) + (w > 1 ? c.y : 0u) + (w > 2 ? c.z : 0u); }
                    runl[bb] += c.x + c.y + c.z + c.w;
                }
                if (valid) dst[start + mine + before_w + rank] = sq[u];
            }
        }
        __syncthreads();   // (run[] and the count sets are rewritten by the next tile)
    }
}

// One launch per level after the binning: the selection (needs the bucket statistics) and the stable scatter (needs
// only the per-tile bucket counts — NOT the chosen split) are independent, so the first `sel_blocks` workgroups
// select while the others scatter.
template <typename T> __global__ __launch_bounds__(256) void k_split(BuildArgs<T> a, int level, uint32_t sel_blocks) {
    if (blockIdx.x == 0 && a.chunk_cnt[0]) {   // the block sums the NEXT level's k_bin adds to (this level reads the other parity)
        uint32_t* nx = a.chunk_cnt[(level + 1) & 1];
        for (uint32_t i = threadIdx.x; i < a.n_chunks * 2u * (uint32_t)NUM_BUCKETS; i += 256) nx[i] = 0u;
    }
    if (blockIdx.x < sel_blocks) select_role<T>(a, level, blockIdx.x, sel_blocks);
    else scatter_role<T>(a, level, blockIdx.x - sel_blocks, gridDim.x - sel_blocks);
}

// ------------------------------------------------------------------------------------------------
// tier 1 in ONE launch per level.  k_level(L) is scatter(L-1) and bin(L) of the two-launch schedule above, fused, with the
// selection of level L-1 recomputed by every workgroup that needs it:
//   a workgroup owns one tile of a level-(L-1) item P.  Wave 0 merges P's statistic replicas and runs the SAH selection
//   (bvh_node.rs:224-247) — every tile of P does, redundantly: the alternative is a launch boundary (≈2.5 µs + the chain of
//   the selection's own round trips) per level.  The other waves meanwhile add up the bucket counts of P's earlier tiles.
//   Then every shape of the tile moves to its place in the stable bucket-major order (:250-272) and — if the child it lands
//   in stays in this tier — is binned for the NEXT split right away: bucket id against the child's centroid bounds
//   (:204-217), 6 x (count, AABB, centroid AABB) per child in LDS (key atomics, 16 replicas), merged into the child's
//   statistics with global atomics; the bucket counts per child tile likewise.  The child's statistics slot and tile numbers
//   are arithmetic (LevelArgs), so no workgroup waits for another.  The workgroup of P's tile 0 also writes P's BvhNode and
//   the children's work items (level tier: LevelArgs arrays; workgroup / wave tier: their queues, as before).
// ROOT: level 0 has no parent — the root item (k_root) is binned in place.
// ------------------------------------------------------------------------------------------------
// Accesses to data another workgroup of the SAME launch wrote (k_level<T, false, true>: one level's result is the next level's input without a
// kernel boundary between them).  DEV: device-scope relaxed atomics = loads / stores with sc1 — a store is written through, a load does not
// hit a line the L1 or a foreign XCD's L2 kept (tools/ubench/twostage.hip: no stale read in 19 M across the chip) — so the result does not
// depend on where the dispatcher put a workgroup.  !DEV: the plain access of the launch-per-level schedule.
template <bool DEV, typename U> __device__ __forceinline__ U ldx(const U* p) {
    if constexpr (DEV) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
#ifndef BVH_XCD_PLAIN_STORES
#define BVH_XCD_PLAIN_STORES 0   // developer variant: 1 = plain stores in the persistent tier too (correct only while group q really sits on one XCD)
#endif
template <bool DEV, typename U, typename V> __device__ __forceinline__ void stx(U* p, V v) {
    if constexpr (DEV && !BVH_XCD_PLAIN_STORES) __hip_atomic_store(p, (U)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = (U)v;
}
template <bool DEV> __device__ __forceinline__ uint4 ldx4(const uint4* p) {
    if constexpr (DEV) { const uint32_t* q = reinterpret_cast<const uint32_t*>(p); return make_uint4(ldx<true>(q), ldx<true>(q + 1), ldx<true>(q + 2), ldx<true>(q + 3)); }
    else return *p;
}
template <bool DEV> __device__ __forceinline__ void stx4(uint4* p, uint4 v) {
    if constexpr (DEV) { uint32_t* q = reinterpret_cast<uint32_t*>(p); stx<true>(q, v.x); stx<true>(q + 1, v.y); stx<true>(q + 2, v.z); stx<true>(q + 3, v.w); }
    else *p = v;
}
__device__ __forceinline__ float lane_bcast_rt(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_bcast_rt(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename T> struct LevelSel {
    uint32_t nl, no_winner;
    T AL[6], CL[6], AR[6], CR[6];
};
// sah_select by ONE WAVE with the work spread over its lanes (the serial form above is ~1 400 instructions on the critical
// path of a level; this is ~200): lanes (b, k) build the running prefix / suffix joins of the AABB keys, lanes 0..4 price the
// five candidate splits — each with exactly the operations of bvh_node.rs:231-238 —, every lane then replays the strict-<
// first-wins scan over the five costs (:239-247), lanes 0..5 assemble the children's bounds.  stk, stc, out, scratch: LDS.
template <typename T>
__device__ __forceinline__ void sah_select_wave(const typename Traits<T>::Key* stk /* 6 x 12 keys */, const uint32_t* stc /* 6 counts */,
                                                const T* A, bool degen, LevelSel<T>* out,
                                                typename Traits<T>::Key* scratch /* 72 keys */, int lane) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    if (lane < 36) {
        const int b = lane / 6, k = lane % 6;
        const bool mn = k < 3;
        Key p = stk[k];
#pragma unroll
        for (int bb = 1; bb < NUM_BUCKETS; bb++) {
            const Key x = stk[bb * STAT_KEYS + k];
            const Key j = mn ? (x < p ? x : p) : (x > p ? x : p);
            p = bb <= b ? j : p;
        }
        Key q = stk[(NUM_BUCKETS - 1) * STAT_KEYS + k];
#pragma unroll
        for (int bb = NUM_BUCKETS - 2; bb >= 0; bb--) {
            const Key x = stk[bb * STAT_KEYS + k];
            const Key j = mn ? (x < q ? x : q) : (x > q ? x : q);
            q = bb >= b ? j : q;
        }
        scratch[b * 6 + k] = p;          // join of buckets 0..b
        scratch[36 + b * 6 + k] = q;     // join of buckets b..5
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t cnt[NUM_BUCKETS], total = 0;
#pragma unroll
    for (int b = 0; b < NUM_BUCKETS; b++) { cnt[b] = stc[b]; total += cnt[b]; }
    T cost = Tr::inf();
    if (lane < NUM_BUCKETS - 1) {
        const int sp = lane;
        uint32_t ln = 0;
#pragma unroll
        for (int b = 0; b < NUM_BUCKETS - 1; b++) ln += b <= sp ? cnt[b] : 0u;
        T la[6], ra[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { la[k] = Tr::unkey(scratch[sp * 6 + k]); ra[k] = Tr::unkey(scratch[36 + (sp + 1) * 6 + k]); }
        const T sa_parent = surface_area(A);
        const T cl = (T)ln * surface_area(la);
        const T cr = (T)(total - ln) * surface_area(ra);
        const T num = cl + cr;
        cost = num / sa_parent;                                  // :236-238
    }
    T min_cost = Tr::inf();
    int best = -1;   // no winner (NaN/inf costs): the reference keeps min_bucket = 0 and EMPTY child bounds (:225-230)
#pragma unroll
    for (int sp = 0; sp < NUM_BUCKETS - 1; sp++) {
        const T c = lane_bcast_rt(cost, sp);   // (v_readlane: no LDS round trip)
        const bool take = degen ? (sp == 0) : (c < min_cost);   // strict <, first wins (:239)
        if (take) { min_cost = c; best = sp; }
    }
    uint32_t nl = 0;
#pragma unroll
    for (int b = 0; b < NUM_BUCKETS; b++) nl += (best < 0 ? b == 0 : b <= best) ? cnt[b] : 0u;
    if (lane < 6) {
        const int k = lane;
        const bool mn = k < 3;
        T al, ar, cl, cr;
        if (best < 0) {
            al = ar = cl = cr = mn ? Tr::inf() : -Tr::inf();
        } else {
            Key kl = mn ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF, kr = kl;
#pragma unroll
            for (int b = 0; b < NUM_BUCKETS; b++) {
                const Key x = stk[b * STAT_KEYS + 6 + k];
                const Key jl = mn ? (x < kl ? x : kl) : (x > kl ? x : kl);
                const Key jr = mn ? (x < kr ? x : kr) : (x > kr ? x : kr);
                kl = b <= best ? jl : kl;
                kr = b <= best ? kr : jr;
            }
            al = Tr::unkey(scratch[best * 6 + k]);
            ar = Tr::unkey(scratch[36 + (best + 1) * 6 + k]);
            cl = Tr::unkey(kl); cr = Tr::unkey(kr);
        }
        out->AL[k] = al; out->AR[k] = ar; out->CL[k] = cl; out->CR[k] = cr;
    }
    if (lane == 0) { out->nl = nl; out->no_winner = best < 0 ? 1u : 0u; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T> struct LevelChild {
    uint32_t start, count, kind, slot, tile0, ax, degen, half;   // kind: 3 level tier, 1 workgroup tier, 0 wave tier
    T cmin, ext;
};
constexpr int LEVEL_TGT = 4;   // child tiles the shapes of ONE (parent tile, bucket) can land in: a run of <= TILE positions
                               // touches <= 2 tiles of a child, and it may straddle the boundary between the two children

template <typename T> __device__ __forceinline__ void level_child_derive(LevelChild<T>* c, const T* C, uint32_t start, uint32_t count,
                                                                          uint32_t mid_max, uint32_t slot_div) {
    c->start = start; c->count = count;
    c->kind = count <= (uint32_t)SMALL_MAX ? 0u : (count <= mid_max ? 1u : 3u);
    c->slot = start / slot_div;
    c->tile0 = start / (uint32_t)TILE + c->slot;
    const int ax = largest_axis(C);                     // bvh_node.rs:107
    c->ax = (uint32_t)ax;
    c->cmin = C[ax];
    c->ext = C[3 + ax] - C[ax];                         // :108
    c->degen = (c->ext < Traits<T>::eps()) ? 1u : 0u;   // :114
    c->half = count / 2;                                // :117
}

#ifdef BVH_LEVEL_PROFILE   // developer build: wall-clock stamps (100 MHz) of k_level's phases at level BVH_LEVEL_PROFILE, per workgroup
__device__ unsigned long long g_level_prof[2 * 8 * 1024];   // rows 0 .. 1023: level BVH_LEVEL_PROFILE, rows 1024 .. 2047: the level after it
#define LEVEL_STAMP(i) do { if ((L == BVH_LEVEL_PROFILE || L == BVH_LEVEL_PROFILE + 1) && threadIdx.x == 0 && blockIdx.x < 1024) \
        g_level_prof[8 * (blockIdx.x + 1024 * (L - BVH_LEVEL_PROFILE)) + (i)] = wall_clock64(); } while (0)
#else
#define LEVEL_STAMP(i) do { } while (0)
#endif
// sum of a u32 over the wave, valid in every lane: 4 DPP row_shr steps (zero for lanes without a source) leave each row's
// sum in its last lane; the four row sums are read by lane number
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    return (uint32_t)(__builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) +
                      __builtin_amdgcn_readlane(v, 63));
}
constexpr int LEVEL_TC_REP = 4;   // LDS replicas of the per-(bucket, target tile) counts
constexpr int LEVEL_PT = TILE / 256;   // shapes per thread

#ifndef BVH_LEVEL_THREADS
#define BVH_LEVEL_THREADS 256
#endif
#ifndef BVH_LEVEL_EARLY_DUTIES
#define BVH_LEVEL_EARLY_DUTIES 1
#endif
// 256: wave 0 runs the selection and then carries shapes like the other three.  320: wave 0 ONLY selects (and writes the node /
// the children's items), four more waves carry the shapes — measured slower (build 0.204 → 0.226 ms at 120 k): five-wave
// workgroups start up to 4 µs apart.
constexpr int LEVEL_THREADS = BVH_LEVEL_THREADS;
constexpr bool LEVEL_DEDICATED = LEVEL_THREADS > 256;
// DEV = false: ONE pass of the tier over all tile ids by the whole grid (a launch per level).
// DEV = true (k_level<T, false, true>, BVHGPU_TUNE_BUILD_LEVEL_PERSIST): the tier's passes from tree level L_first on as ONE
// persistent launch.  The launch-per-level schedule pays ≈ 2.5 µs of dispatch gap + ≈ 2 µs of cold misses per level; a barrier over ALL
// workgroups costs as much (2.3 – 2.5 µs in two stages, tools/ubench/twostage.hip), but the eight subtrees below tree level 3 never exchange
// anything: workgroup group q = blockIdx.x % 8 (the dispatcher puts those on one XCD: 2 048 of 2 048 blocks — a matter of speed only, see
// ldx / stx) owns the subtree of heap number 8 + q and synchronises with ITSELF after every pass (1.9 µs for 32 workgroups, eight groups
// side by side), running on until its subtree has left the tier, however deep.  A pass covers the subtree's tile ids [T(start), T(start +
// count)), T(p) = p / TILE + p / slot_div (monotone in p: the ranges of the eight subtrees do not overlap, LevelArgs) and resets the next
// pass's accumulators over that range and the statistics slots [start / slot_div, end / slot_div) only.  Barrier: one 64-bit word per group,
// {arrivals | children sent on << 32}; the last arriver publishes {round | children << 32} on a line of its own, the others poll that with
// sc1 loads.  The release is s_waitcnt vmcnt(0) behind write-through stores: no fence (an agent-scope fence writes the whole L2 back:
// 6.5 µs).  A poll that does not come back (the group's workgroups are not all resident: another stream's walk holds the CUs) raises
// BUILD_FLAG_PERSIST_GAVE_UP and the host builds the tree again with a launch per level (build_finalize).
// (One kernel body for both: `a` must stay the kernel's own by-value parameter — handed to a helper by reference it is copied to scratch,
//  344 bytes, and every access to it becomes a scratch load: 182 VGPRs and +67 µs per build, measured.)
constexpr unsigned long long XCD_SPIN_TICKS = 200000ull;   // 2 ms of the 100 MHz wall clock: a group barrier that has not come back by then never will (ADVICE r5:
                                                           // the bound used to be a poll COUNT — seconds of a stalled stream when the group's workgroups are not co-resident)
template <typename T, bool ROOT, bool DEV = false> __global__ __launch_bounds__(LEVEL_THREADS) void k_level(BuildArgs<T> a, int L_first) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    static_assert(TILE % 256 == 0 && LEVEL_PT >= 1 && LEVEL_PT <= 4, "a tile is a whole number of 256-thread rounds");
    static_assert(!(ROOT && DEV), "the root is binned by a launch of its own");
    const LevelArgs<T>& v = a.lv;
    uint32_t wg_rank = blockIdx.x, wg_count = gridDim.x, g0 = 0u, g1 = v.n_tiles, s0 = 0u, s1 = v.n_slots;
    __shared__ uint32_t s_live, s_go;
    __shared__ unsigned long long s_word;
    uint32_t live_seen = 0;
    if constexpr (DEV) {
        const uint32_t grp = blockIdx.x & 7u;
        wg_rank = blockIdx.x >> 3; wg_count = gridDim.x >> 3;
        const uint32_t* xd = &a.ctr[CTR_XDIR + 4 * grp];        // (written by the pass that split tree level 2: an earlier launch)
        const uint32_t kind = xd[0], start = xd[2], count = xd[3];
        if (kind != 3u || (a.ctr[CTR_FLAGS] & BUILD_FLAG_NONFINITE)) return;   // the whole group: its subtree never reached this tier (or there is no tree)
        const uint32_t sd = v.slot_div, end = start + count;
        g0 = start / (uint32_t)TILE + start / sd; g1 = end / (uint32_t)TILE + end / sd;
        s0 = start / sd; s1 = end / sd;
    }
    uint32_t* const live = &s_live;
  for (int L = L_first, round = 0; ; L++, round++) {
    if constexpr (DEV) {
        if (threadIdx.x == 0) s_live = 0u;
        __syncthreads();
    }
    const int bP = (L + 2) % 3, bC = L % 3, bN = (L + 1) % 3;
    const uint32_t* src = ROOT ? a.idx[0] : a.idx[(L + 1) & 1];   // the parents' order (level L-1; the root's slice is where k_prep wrote it)
    uint32_t* dst = a.idx[L & 1];                                   // the children's
    const uint8_t* bk_src = v.bk[(L + 1) & 1];
    uint8_t* bk_dst = v.bk[L & 1];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const bool shaper = !LEVEL_DEDICATED || w > 0;                            // this thread carries shapes
    const int sw = LEVEL_DEDICATED ? w - 1 : w;                               // its wave among the shape waves
    const uint32_t stid = LEVEL_DEDICATED ? threadIdx.x - 64u : threadIdx.x;   // its number among the 256 shape threads
    const unsigned long long lt = lanemask_lt();
    LEVEL_STAMP(0);
    // (the first tile's record is requested before the housekeeping stores: its round trip hides behind them)
    uint4 tm_first = make_uint4(NONE, 0u, 0u, 0u);
    if (!ROOT && g0 + wg_rank < g1) tm_first = ldx4<DEV>(&v.tile_map[bP][g0 + wg_rank]);
    // ROOT: the scene bounds (k_prep left one row of 12 keys per workgroup) → the root item, in every workgroup; tile g of
    // the root is simply positions [g TILE, (g+1) TILE)
    __shared__ Key s_rootk[STAT_KEYS];
    if (ROOT) {
        if (threadIdx.x < STAT_KEYS) s_rootk[threadIdx.x] = key_is_min(threadIdx.x) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < a.prep_wgs * (uint32_t)STAT_KEYS; e += LEVEL_THREADS) {   // rows x keys, coalesced
            const Key x = a.rootkeys[e];
            const int j = (int)(e % (uint32_t)STAT_KEYS);
            if (key_is_min(j)) atomicMin(&s_rootk[j], x);
            else atomicMax(&s_rootk[j], x);
        }
        __syncthreads();
        bool bad = (a.ctr[CTR_FLAGS] & BUILD_FLAG_NONFINITE) != 0u;
        // finite boxes whose centroid extent overflows: (c - cmin) / ext is NaN for some shape → the reference's panic (k_root)
#pragma unroll
        for (int k = 0; k < 3; k++) bad = bad || !(fabs(Tr::unkey(s_rootk[9 + k]) - Tr::unkey(s_rootk[6 + k])) < Tr::inf());
        if (bad) {   // invalid input: nothing is queued, every later kernel of the optimistic schedule finds nothing to do
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&a.ctr[CTR_FLAGS], BUILD_FLAG_NONFINITE);
            return;
        }
        const uint32_t root_tiles = (a.n + TILE - 1) / TILE;
        if (blockIdx.x == 0) {   // the record and tile map the next level reads
            Item<T>* it = &v.item[0][0];
            if (threadIdx.x == 0) {
                it->ni = 0; it->parent = 0; it->start = 0; it->count = a.n; it->tile_base = 0; it->parity = 0; it->heap = 1u; it->_r1 = 0;
                a.ctr[CTR_LEVEL0] = 1u;
            }
            if (threadIdx.x < 6) { it->A[threadIdx.x] = Tr::unkey(s_rootk[threadIdx.x]); it->C[threadIdx.x] = Tr::unkey(s_rootk[6 + threadIdx.x]); }
            for (uint32_t j = threadIdx.x; j < root_tiles; j += LEVEL_THREADS) v.tile_map[0][j] = make_uint4(0u, 0u, a.n, 0u);
        }
        if (blockIdx.x < root_tiles) tm_first = make_uint4(0u, 0u, a.n, 0u);
    }

    // ---- reset what the NEXT level accumulates into
    {
        const uint32_t gt = wg_rank * (uint32_t)LEVEL_THREADS + threadIdx.x, gs = wg_count * (uint32_t)LEVEL_THREADS;
        for (uint32_t i = g0 + gt; i < g1; i += gs) {
            if constexpr (DEV) stx<true>(&v.tile_map[bN][i].x, NONE);   // (readers look at .x first)
            else v.tile_map[bN][i] = make_uint4(NONE, 0u, 0u, 0u);
        }
        for (uint32_t i = g0 * (uint32_t)NUM_BUCKETS + gt; i < g1 * (uint32_t)NUM_BUCKETS; i += gs) stx<DEV>(&v.tile_cnt[bN][i], 0u);
        constexpr uint32_t NK = NUM_BUCKETS * STAT_KEYS;
        const uint32_t e0 = s0 * (uint32_t)STAT_REP, e1 = s1 * (uint32_t)STAT_REP;
        for (uint32_t i = e0 * NK + gt; i < e1 * NK; i += gs) {
            const uint32_t e = i / NK, j = i % NK;
            stx<DEV>(&v.stats[bN][e].k[j], key_is_min((int)(j % STAT_KEYS)) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF);
        }
        for (uint32_t i = e0 * (uint32_t)NUM_BUCKETS + gt; i < e1 * (uint32_t)NUM_BUCKETS; i += gs) stx<DEV>(&v.stats[bN][i / NUM_BUCKETS].cnt[i % NUM_BUCKETS], 0u);
    }

    __shared__ Key sk[2][BIN_REP][NUM_BUCKETS * STAT_KEYS];
    __shared__ uint32_t sc[2][BIN_REP][NUM_BUCKETS];
    __shared__ uint32_t tcnt[LEVEL_TC_REP][NUM_BUCKETS][LEVEL_TGT][NUM_BUCKETS];   // [replica][bucket at L-1][target tile][bucket at L]
    __shared__ uint32_t run0[NUM_BUCKETS], wsum[4][NUM_BUCKETS], wcnt[4][LEVEL_PT][NUM_BUCKETS];
    __shared__ LevelChild<T> ch[2];
    __shared__ LevelSel<T> sel;
    __shared__ ItemStats<T> s_merged;
    __shared__ Key s_sah[72];
    const int rp = (int)(stid & (BIN_REP - 1)), rt = (int)(stid & (LEVEL_TC_REP - 1));

    for (uint32_t g = g0 + wg_rank; g < g1; g += wg_count) {
        // one load tells the tile's workgroup all it needs to start: the item's slot (statistics, record), slice and size
        const uint4 tm = g == g0 + wg_rank ? tm_first : (ROOT ? make_uint4(g < (a.n + TILE - 1) / TILE ? 0u : NONE, 0u, a.n, 0u) : ldx4<DEV>(&v.tile_map[bP][g]));
        const uint32_t slotP = tm.x;
        if (slotP == NONE) continue;   // workgroup-uniform
        LEVEL_STAMP(1);
        const uint32_t start = tm.y, count = tm.z;
        const Item<T>* P = &v.item[ROOT ? 0 : ((L + 1) & 1)][slotP];
        const uint32_t tile0P = start / (uint32_t)TILE + slotP;
        const uint32_t tl = g - tile0P, ntl = (count + TILE - 1) / TILE;
        const uint32_t p0 = start + tl * TILE;
        const uint32_t pend = min(start + count, p0 + (uint32_t)TILE);

        // ---- wave 0 starts with what the selection waits for (P's statistic replicas: lane j owns keys j, j + 64; P's bounds):
        // memory returns in order, so these must be in flight before anything else this wave asks for
        Key rk[2][STAT_REP];
        uint32_t rc[STAT_REP];
        T PA[6], PC[6];
        uint32_t P_ni = 0, P_heap = 0, P_parent = 0;   // (tile 0's duties: the same record, fetched with the bounds)
        if (!ROOT && w == 0) {
            const ItemStats<T>* rep = &v.stats[bP][(size_t)slotP * STAT_REP];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = lane + 64 * h;
#pragma unroll
                for (int r = 0; r < STAT_REP; r++) rk[h][r] = j < NUM_BUCKETS * STAT_KEYS ? ldx<DEV>(&rep[r].k[j]) : (Key)0;
            }
#pragma unroll
            for (int r = 0; r < STAT_REP; r++) rc[r] = lane < NUM_BUCKETS ? ldx<DEV>(&rep[r].cnt[lane]) : 0u;
#pragma unroll
            for (int k = 0; k < 6; k++) { PA[k] = ldx<DEV>(&P->A[k]); PC[k] = ldx<DEV>(&P->C[k]); }
            if (tl == 0) { P_ni = ldx<DEV>(&P->ni); P_heap = ldx<DEV>(&P->heap); P_parent = ldx<DEV>(&P->parent); }
        }
        // ---- loads that depend on the tile only: the shapes' indices and buckets at L-1 ...
        uint32_t sh[LEVEL_PT];
        int bo[LEVEL_PT];
        T bx[LEVEL_PT][6];
#pragma unroll
        for (int u = 0; u < LEVEL_PT; u++) {
            const uint32_t p = p0 + (uint32_t)u * 256u + stid;
            const bool valid = shaper && p < pend;
            sh[u] = valid ? ldx<DEV>(&src[p]) : NONE;
            bo[u] = !valid ? 7 : (ROOT ? 0 : (int)ldx<DEV>(&bk_src[p]));
        }
        // ... and the bucket counts of P's tiles (consumed further down) → offset of (this tile, bucket b) inside P's slice =
        // shapes of P in buckets < b + shapes of bucket b in P's earlier tiles
        constexpr int TCV = 2;   // tiles per thread held in registers before they are consumed (more are fetched in a loop)
        constexpr int NCT = LEVEL_THREADS - 64;             // the selection wave takes no part in this
        const uint32_t ctid = threadIdx.x - 64u;
        uint32_t tcv[TCV][NUM_BUCKETS];
        const uint32_t* tc = v.tile_cnt[bP] + (size_t)tile0P * NUM_BUCKETS;
        if (!ROOT) {
#pragma unroll
            for (int i = 0; i < TCV; i++) {
                const uint32_t j = ctid + (uint32_t)NCT * (uint32_t)i;
#pragma unroll
                for (int b = 0; b < NUM_BUCKETS; b++) tcv[i][b] = (w > 0 && j < ntl) ? ldx<DEV>(&tc[(size_t)j * NUM_BUCKETS + b]) : 0u;
            }
        }
        // ---- tile 0 of P: P's BvhNode (bvh_node.rs:145-151) and the children's work items.  Everything it needs is known once the selection
        // is: wave 0 issues these stores right behind the selection (BVH_LEVEL_EARLY_DUTIES, default), where their latency hides behind the
        // move of the shapes — at the end of the pass (the round-2 .. 4 place) the tile-0 workgroups were the stragglers every launch boundary
        // and every group barrier waited for: mean end of a pass 7.6 µs, last end 9.7 µs (tools/level_prof.py, profiles/r5_level_timeline.log)
        auto tile0_duties = [&](const uint32_t nl) {
            const uint32_t ni = P_ni, heap = P_heap;
            const uint32_t li = ni + 1;                 // :140
            const uint32_t ri = li + (2 * nl - 1);      // :138,142 (nl: the lambda's argument)
            if (sel.no_winner && lane == 0) atomicOr(&a.ctr[CTR_FLAGS], BUILD_FLAG_EMPTY_SPLIT);
            if (lane == 0) {
                typename Tr::Node* nd = &a.nodes[ni];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    nd->l_min[k] = sel.AL[k]; nd->l_max[k] = sel.AL[3 + k];
                    nd->r_min[k] = sel.AR[k]; nd->r_max[k] = sel.AR[3 + k];
                }
                nd->parent = P_parent; nd->l = li; nd->r = ri; nd->shape = NONE;
                a.node_start[ni] = start;
                a.node_count[ni] = count;
                a.node_slot[ni] = (uint16_t)heap;
                if (heap < 16u) atomicOr(&a.ctr[CTR_TOPMASK], 1u << heap);
            }
            // queue slots of the children that leave this tier: lanes 0 / 1 reserve them at the same time.  Only then — the tier's last one or
            // two levels — does this wave wait for anything: the slot comes back from a returning atomic, and the wait for it is a wait for
            // EVERYTHING the wave has in flight (the compiler can only count, s_waitcnt vmcnt(0)), the node record stored above included: with
            // the wait on every level's path the first-tile workgroups spent 2.0 – 2.3 µs here against 0.7 for the others and were the last to
            // finish every pass (tools/level_prof.py).  Children that stay in the tier need no slot: nothing to wait for.
            const uint32_t mykind = ch[lane & 1].kind;
            uint32_t cqs[2] = {0u, 0u};
            if (ch[0].kind != 3u || ch[1].kind != 3u) {   // (wave-uniform)
                uint32_t qslot = 0;
                if (lane < 2) {
                    if (mykind == 0u) qslot = atomicAdd(&a.ctr[CTR_SMALL], 1u);
                    else if (mykind == 1u) qslot = atomicAdd(&a.ctr[CTR_MID2], 1u);
                }
                cqs[0] = __shfl(qslot, 0); cqs[1] = __shfl(qslot, 1);
            }
            if (lane < 2 && mykind == 3u) atomicAdd(&a.ctr[CTR_LEVEL0 + 2 * lvl_slot(L)], 1u);   // the host only asks whether the level is empty
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const LevelChild<T>& c = ch[side];
                const uint32_t cq = cqs[side];
                Item<T>* it = c.kind == 0u ? &a.small[cq] : (c.kind == 1u ? &a.mid2[cq] : &v.item[L & 1][c.slot]);
                if (lane == 0) {
                    stx<DEV>(&it->ni, side ? ri : li); stx<DEV>(&it->parent, ni); stx<DEV>(&it->start, c.start); stx<DEV>(&it->count, c.count);
                    stx<DEV>(&it->tile_base, c.tile0); stx<DEV>(&it->parity, (uint32_t)(L & 1)); stx<DEV>(&it->heap, heap_child(heap, (uint32_t)side)); stx<DEV>(&it->_r1, 0u);
                    // the tree's level-3 nodes (heap numbers 8 .. 15): the subtrees the groups of k_level<T, false, true> own
                    const uint32_t hc = heap_child(heap, (uint32_t)side);
                    if (hc >= 8u && hc < 16u) {
                        uint32_t* xd = &a.ctr[CTR_XDIR + 4 * (hc - 8u)];
                        xd[0] = c.kind; xd[1] = c.slot; xd[2] = c.start; xd[3] = c.count;
                    }
                    if (DEV && c.kind == 3u) atomicAdd(live, 1u);   // (LDS)
                }
                if (lane < 6) { stx<DEV>(&it->A[lane], side ? sel.AR[lane] : sel.AL[lane]); stx<DEV>(&it->C[lane], side ? sel.CR[lane] : sel.CL[lane]); }
                if (c.kind == 3u) {
                    const uint32_t cnt_t = (c.count + TILE - 1) / TILE;
                    for (uint32_t j = lane; j < cnt_t; j += WAVE) stx4<DEV>(&v.tile_map[bC][c.tile0 + j], make_uint4(c.slot, c.start, c.count, 0u));
                }
            }
                };
        if (ROOT) {
            if (threadIdx.x == 0) {
                T C[6];
#pragma unroll
                for (int k = 0; k < 6; k++) C[k] = Tr::unkey(s_rootk[6 + k]);
                level_child_derive<T>(&ch[0], C, start, count, a.mid_max, v.slot_div);
                ch[1].start = start + count; ch[1].count = 0; ch[1].kind = 0; ch[1].slot = 0; ch[1].tile0 = 0;
                sel.nl = count; sel.no_winner = 0;
            }
            if (threadIdx.x < NUM_BUCKETS) run0[threadIdx.x] = 0;
        } else if (w == 0) {
            // the selection of P (every tile's workgroup repeats it): replicas merged (joins are exact, counts are integers).
            // Wave 0 does this FIRST — its other loads are in flight, everybody else waits for the outcome.
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = lane + 64 * h;
                if (j < NUM_BUCKETS * STAT_KEYS) {
                    const bool mn = key_is_min(j % STAT_KEYS);
                    Key x = rk[h][0];
#pragma unroll
                    for (int r = 1; r < STAT_REP; r++) { const Key u = rk[h][r]; x = mn ? (u < x ? u : x) : (u > x ? u : x); }
                    s_merged.k[j] = x;
                }
            }
            if (lane < NUM_BUCKETS) {
                uint32_t c = 0;
#pragma unroll
                for (int r = 0; r < STAT_REP; r++) c += rc[r];
                s_merged.cnt[lane] = c;
            }
            const int ax = largest_axis(PC);
            const bool degen = (PC[3 + ax] - PC[ax]) < Tr::eps();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            LEVEL_STAMP(7);
            sah_select_wave<T>(s_merged.k, s_merged.cnt, PA, degen, &sel, s_sah, lane);
#ifdef BVH_LEVEL_PROFILE
            LEVEL_STAMP(3);   // (slot 3: selection done)
#endif
            if (lane < 2) {   // lane 0: left child, lane 1: right child
                const uint32_t nl = sel.nl;
                T CC[6];
#pragma unroll
                for (int k = 0; k < 6; k++) CC[k] = lane ? sel.CR[k] : sel.CL[k];
                level_child_derive<T>(&ch[lane], CC, lane ? start + nl : start, lane ? count - nl : nl, a.mid_max, v.slot_div);
            }
            if (BVH_LEVEL_EARLY_DUTIES && tl == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                tile0_duties(sel.nl);
            }
        } else {
            // the other waves prepare the LDS accumulators meanwhile
            constexpr int NT3 = LEVEL_THREADS - 64;
            const int t3 = (int)threadIdx.x - 64;
            for (int j = t3; j < 2 * BIN_REP * NUM_BUCKETS * STAT_KEYS; j += NT3)
                (&sk[0][0][0])[j] = key_is_min(j % STAT_KEYS) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
            for (int j = t3; j < 2 * BIN_REP * NUM_BUCKETS; j += NT3) (&sc[0][0][0])[j] = 0u;
            for (int j = t3; j < LEVEL_TC_REP * NUM_BUCKETS * LEVEL_TGT * NUM_BUCKETS; j += NT3) (&tcnt[0][0][0][0])[j] = 0u;
        }
        if (ROOT) {   // (no selection: every wave shares the preparation)
            for (int j = threadIdx.x; j < 2 * BIN_REP * NUM_BUCKETS * STAT_KEYS; j += LEVEL_THREADS)
                (&sk[0][0][0])[j] = key_is_min(j % STAT_KEYS) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
            for (int j = threadIdx.x; j < 2 * BIN_REP * NUM_BUCKETS; j += LEVEL_THREADS) (&sc[0][0][0])[j] = 0u;
            for (int j = threadIdx.x; j < LEVEL_TC_REP * NUM_BUCKETS * LEVEL_TGT * NUM_BUCKETS; j += LEVEL_THREADS) (&tcnt[0][0][0][0])[j] = 0u;
        }
        // the shapes' AABBs (their indices have arrived by now)
#pragma unroll
        for (int u = 0; u < LEVEL_PT; u++) {
            if (sh[u] != NONE) {
                const T* bp = a.aabbs + 6 * (size_t)sh[u];
#pragma unroll
                for (int k = 0; k < 6; k++) bx[u][k] = bp[k];
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) bx[u][k] = (T)0;
            }
        }
        if (!ROOT && w > 0) {
            uint32_t before[NUM_BUCKETS], all[NUM_BUCKETS];
#pragma unroll
            for (int b = 0; b < NUM_BUCKETS; b++) { before[b] = 0; all[b] = 0; }
#pragma unroll
            for (int i = 0; i < TCV; i++) {
                const uint32_t j = ctid + (uint32_t)NCT * (uint32_t)i;
#pragma unroll
                for (int b = 0; b < NUM_BUCKETS; b++) { all[b] += tcv[i][b]; before[b] += j < tl ? tcv[i][b] : 0u; }
            }
            for (uint32_t j = ctid + (uint32_t)NCT * TCV; j < ntl; j += NCT) {   // (items of more than 384 tiles: huge scenes' top levels)
#pragma unroll
                for (int b = 0; b < NUM_BUCKETS; b++) {
                    const uint32_t x = ldx<DEV>(&tc[(size_t)j * NUM_BUCKETS + b]);
                    all[b] += x;
                    before[b] += j < tl ? x : 0u;
                }
            }
            uint32_t acc = 0;
#pragma unroll
            for (int b = 0; b < NUM_BUCKETS; b++) {
                const uint32_t mine = wave_sum_u32(acc + before[b]);
                acc += all[b];
                if (lane == 0) wsum[w - 1][b] = mine;
            }
        }
        // ranks inside the tile: ballots per 256-shape round, counts per (wave, round, bucket) through LDS
        uint32_t rank[LEVEL_PT];
#pragma unroll
        for (int u = 0; u < LEVEL_PT; u++) rank[u] = 0;
        if (!ROOT && shaper) {
#pragma unroll
            for (int u = 0; u < LEVEL_PT; u++) {
                rank[u] = 0;
#pragma unroll
                for (int bb = 0; bb < NUM_BUCKETS; bb++) {
                    const unsigned long long m = __ballot(bo[u] == bb);
                    if (bo[u] == bb) rank[u] = (uint32_t)__popcll(m & lt);
                    if (lane == 0) wcnt[sw][u][bb] = (uint32_t)__popcll(m);
                }
            }
        }
        LEVEL_STAMP(2);
        lds_barrier();
        if (!ROOT && threadIdx.x < NUM_BUCKETS) {
            uint32_t r0 = 0;
#pragma unroll
            for (int ww = 0; ww < LEVEL_THREADS / 64 - 1; ww++) r0 += wsum[ww][threadIdx.x];
            run0[threadIdx.x] = r0;
        }
        if (!ROOT) lds_barrier();
        const uint32_t nl = sel.nl;
        // ---- the tile's shapes: stable bucket-major move (bvh_node.rs:250-272) + bucket / statistics for the next split
#pragma unroll
        for (int u = 0; u < LEVEL_PT; u++) {
            if (sh[u] == NONE) continue;
            const uint32_t p = p0 + (uint32_t)u * 256u + stid;
            const int b = bo[u];
            uint32_t off;
            if (ROOT) {
                off = p - start;
            } else {
                off = run0[b] + rank[u];
                for (int uu = 0; uu < u; uu++) off += wcnt[0][uu][b] + wcnt[1][uu][b] + wcnt[2][uu][b] + wcnt[3][uu][b];   // earlier rounds
                for (int ww = 0; ww < sw; ww++) off += wcnt[ww][u][b];                                                     // earlier waves of this round
                stx<DEV>(&dst[start + off], sh[u]);
            }
            const int side = off >= nl ? 1 : 0;
            const LevelChild<T>& c = ch[side];
            if (c.kind == 3u) {
                T cen[3];
#pragma unroll
                for (int k = 0; k < 3; k++) cen[k] = center1(bx[u][k], bx[u][3 + k]);
                const uint32_t q = start + off;
                int nb;
                if (c.degen) nb = (q - c.start) < c.half ? 0 : 1;          // halves in the child's order (:117)
                else nb = bucket_of(cen[c.ax], c.cmin, c.ext);             // :210-217
                stx<DEV>(&bk_dst[q], (uint8_t)nb);
                Key* kk = &sk[side][rp][nb * STAT_KEYS];                   // Bucket::add_aabb (utils.rs:81-85)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    atomicMin(&kk[k], Tr::key(bx[u][k]));
                    atomicMax(&kk[3 + k], Tr::key(bx[u][3 + k]));
                    const Key kc = Tr::key(cen[k]);
                    atomicMin(&kk[6 + k], kc);
                    atomicMax(&kk[9 + k], kc);
                }
                atomicAdd(&sc[side][rp][nb], 1u);
                // which of the (at most LEVEL_TGT) child tiles this (parent tile, bucket) run reaches: relative to the run's first shape
                const uint32_t off_first = ROOT ? (p0 - start) : run0[b];
                const int side_first = off_first >= nl ? 1 : 0;
                const uint32_t kt = (q - c.start) / (uint32_t)TILE;
                const uint32_t kt_first = (start + off_first - ch[side_first].start) / (uint32_t)TILE;
                const uint32_t tgt = side != side_first ? 2u + kt : kt - kt_first;
                atomicAdd(&tcnt[rt][b][tgt][nb], 1u);
            }
        }
        lds_barrier();
        LEVEL_STAMP(4);
        // ---- merge into the children's statistics (replica = this tile's number in P, like k_bin) and tile counts
        for (int e = threadIdx.x; e < 2 * NUM_BUCKETS * STAT_KEYS; e += LEVEL_THREADS) {
            const int side = e / (NUM_BUCKETS * STAT_KEYS), j = e % (NUM_BUCKETS * STAT_KEYS);
            if (ch[side].kind != 3u) continue;
            uint32_t cn = 0;
#pragma unroll
            for (int r = 0; r < BIN_REP; r++) cn += sc[side][r][j / STAT_KEYS];
            if (!cn) continue;
            Key x = sk[side][0][j];
            const bool mn = key_is_min(j % STAT_KEYS);
#pragma unroll
            for (int r = 1; r < BIN_REP; r++) { const Key u = sk[side][r][j]; x = mn ? (u < x ? u : x) : (u > x ? u : x); }
            ItemStats<T>* gs = &v.stats[bC][(size_t)ch[side].slot * STAT_REP + (tl & (STAT_REP - 1))];
            if (mn) atomicMin(&gs->k[j], x);
            else atomicMax(&gs->k[j], x);
            if (j % STAT_KEYS == 0) atomicAdd(&gs->cnt[j / STAT_KEYS], cn);
        }
        for (int e = threadIdx.x; e < NUM_BUCKETS * LEVEL_TGT * NUM_BUCKETS; e += LEVEL_THREADS) {
            const int b = e / (LEVEL_TGT * NUM_BUCKETS), tgt = (e / NUM_BUCKETS) % LEVEL_TGT, nb = e % NUM_BUCKETS;
            uint32_t cn = 0;
#pragma unroll
            for (int r = 0; r < LEVEL_TC_REP; r++) cn += tcnt[r][b][tgt][nb];
            if (!cn) continue;
            const uint32_t off_first = ROOT ? (p0 - start) : run0[b];
            const int side_first = off_first >= nl ? 1 : 0;
            const int side = tgt >= 2 ? 1 : side_first;
            const uint32_t kt_first = (start + off_first - ch[side_first].start) / (uint32_t)TILE;
            const uint32_t kt = tgt >= 2 ? (uint32_t)tgt - 2u : kt_first + (uint32_t)tgt;
            atomicAdd(&v.tile_cnt[bC][(size_t)(ch[side].tile0 + kt) * NUM_BUCKETS + nb], cn);
        }
        LEVEL_STAMP(5);
        if (!BVH_LEVEL_EARLY_DUTIES && !ROOT && tl == 0 && w == 0) tile0_duties(nl);
        LEVEL_STAMP(6);
        lds_barrier();
    }
    if constexpr (!DEV) {
        break;
    } else {
        // ---- the group's barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every store and atomic of this pass has been performed
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long* arrive = a.xbar + (size_t)(blockIdx.x & 7u) * 32, *go = arrive + 16;
            const unsigned long long old = __hip_atomic_fetch_add(arrive, 1ull | ((unsigned long long)s_live << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long word;
            uint32_t ok = 1u;
            if ((uint32_t)old + 1u == (uint32_t)(round + 1) * wg_count) {      // the last one of the group
                word = (unsigned long long)(uint32_t)(round + 1) | (((old >> 32) + (unsigned long long)s_live) << 32);
                __hip_atomic_store(go, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const unsigned long long t0 = wall_clock64();
                bool late = false;
                while ((uint32_t)(word = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (uint32_t)(round + 1) &&
                       !(late = wall_clock64() - t0 > XCD_SPIN_TICKS))
                    __builtin_amdgcn_s_sleep(1);
                if (late) ok = 0u;
            }
            s_word = word; s_go = ok;
        }
        __syncthreads();
        if (!s_go) { if (threadIdx.x == 0) atomicOr(&a.ctr[CTR_FLAGS], BUILD_FLAG_PERSIST_GAVE_UP); return; }
        const uint32_t live_total = (uint32_t)(s_word >> 32);
        if (live_total == live_seen) return;                 // nothing of this subtree stays in the tier: done
        live_seen = live_total;
        if (L + 1 >= MAXLV - 4) { if (threadIdx.x == 0) atomicOr(&a.ctr[CTR_FLAGS], BUILD_FLAG_PERSIST_GAVE_UP); return; }   // (deeper than the counter slots)
    }
  }
}
#ifdef BVH_LEVEL_PROFILE
void debug_level_prof(unsigned long long* out, size_t n) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_level_prof), sizeof(unsigned long long) * n);
}
#endif

// ------------------------------------------------------------------------------------------------
// Wave-level SEGMENTED inclusive scans of the 12 bound values (aabb min3 max3, centroid min3 max3) with join as the
// operator: lane i ends with the join over [lo_i, i] (prefix) / [i, hi_i) (suffix), where [lo_i, hi_i) is the run of
// lanes lane i belongs to.  Inside a row of 16 lanes the partners come by DPP (row_shr / row_shl: a VALU move, no
// LDS crossbar trip — 288 ds_bpermute per level made the wave tier LDS-bound); across rows the carry is one lane's
// value: row_bcast:15 / :31 upwards, v_readlane of lanes 48 / 32 / 16 downwards.  A lane only ever joins a partner of
// its own run.  `span`: an upper bound (power of two) on the length of the live runs: the steps beyond are skipped.
// ------------------------------------------------------------------------------------------------
// N values, in groups of six: three minima, three maxima (N = 12: the statistic keys' order)
template <typename T, int N> __device__ __forceinline__ void join12(T (&a)[N], const T (&b)[N]) {
    static_assert(N % 6 == 0, "min3 max3 groups");
#pragma unroll
    for (int k = 0; k < N; k++) a[k] = (k % 6) < 3 ? join_min(a[k], b[k]) : join_max(a[k], b[k]);
}
template <typename T, int D, int N> __device__ __forceinline__ void seg_prefix_step(T (&P)[N], int lane, int lo) {
    T u[N];
#pragma unroll
    for (int k = 0; k < N; k++) u[k] = dpp_fetch_raw<0x110 + D>(P[k]);
    if (lane - D >= lo && (lane & 15) >= D) join12<T, N>(P, u);   // (source inside the run and inside the row)
}
template <typename T, int D, int N> __device__ __forceinline__ void seg_suffix_step(T (&S)[N], int lane, int hi) {
    T u[N];
#pragma unroll
    for (int k = 0; k < N; k++) u[k] = dpp_fetch_raw<0x100 + D>(S[k]);
    if (lane + D < hi && (lane & 15) + D < 16) join12<T, N>(S, u);
}
template <typename T, int N> __device__ __forceinline__ void seg_prefix_scan(T (&P)[N], int lane, int lo, int span) {
    if (span > 1) seg_prefix_step<T, 1, N>(P, lane, lo);
    if (span > 2) seg_prefix_step<T, 2, N>(P, lane, lo);
    if (span > 4) seg_prefix_step<T, 4, N>(P, lane, lo);
    if (span > 8) seg_prefix_step<T, 8, N>(P, lane, lo);
    const int row0 = lane & ~15;   // first lane of this lane's row
    if (__any(lo < row0)) {        // a run that began in an earlier row takes the finished prefix of the row below
        T u[N];
#pragma unroll
        for (int k = 0; k < N; k++) u[k] = dpp_fetch_raw<0x142, 0xA>(P[k]);   // rows 1 and 3 from lanes 15 / 47
        if ((lane & 16) && lo < row0) join12<T, N>(P, u);
#pragma unroll
        for (int k = 0; k < N; k++) u[k] = dpp_fetch_raw<0x143, 0xC>(P[k]);   // rows 2 and 3 from lane 31 (complete by now)
        if (lane >= 32 && lo < 32) join12<T, N>(P, u);
    }
}
template <typename T, int ROW, int N> __device__ __forceinline__ void seg_suffix_carry(T (&S)[N], int lane, int hi) {
    T u[N];
#pragma unroll
    for (int k = 0; k < N; k++) u[k] = lane_bcast<16 * (ROW + 1)>(S[k]);
    if ((lane & ~15) == 16 * ROW && hi > 16 * (ROW + 1)) join12<T, N>(S, u);
}
template <typename T, int N> __device__ __forceinline__ void seg_suffix_scan(T (&S)[N], int lane, int hi, int span) {
    if (span > 1) seg_suffix_step<T, 1, N>(S, lane, hi);
    if (span > 2) seg_suffix_step<T, 2, N>(S, lane, hi);
    if (span > 4) seg_suffix_step<T, 4, N>(S, lane, hi);
    if (span > 8) seg_suffix_step<T, 8, N>(S, lane, hi);
    if (__any(hi > (lane & ~15) + 16)) {   // top row first: a run that continues into the next row takes that row's first lane
        seg_suffix_carry<T, 2, N>(S, lane, hi);
        seg_suffix_carry<T, 1, N>(S, lane, hi);
        seg_suffix_carry<T, 0, N>(S, lane, hi);
    }
}

// ------------------------------------------------------------------------------------------------
// mid tier — one workgroup finishes a whole node of 65 .. Cfg::MAXN shapes down to <= 64-shape
// sub-nodes, level by level, entirely in LDS: the node's index slice AND its shapes' AABBs are
// loaded once and then only permuted in LDS.  Same-address LDS atomics were the bottleneck of a
// first version (13 per shape per level), so statistics are taken AFTER the stable sort, where every
// (sub-node, bucket) is one contiguous run:
//   1 bucket   : every thread owns PPT consecutive positions: bucket id per shape (bvh_node.rs:204-217)
//   2 sort     : block-wide exclusive scan of packed one-hot bucket counters → stable bucket-major rank
//                inside the sub-node (:250-272); shapes (index + AABB) move through registers
//   3 stats    : thread-serial join over its positions, then a wave-level SEGMENTED scan (head flags)
//                over the 64 per-thread partials; only run tails touch the LDS statistics (key atomics,
//                practically conflict-free)
//   4 select   : lane s of wave 0 runs the SAH selection of sub-node s (:224-247), writes its BvhNode,
//                creates next-level sub-nodes; <= 64-shape children go to the global wave-tier queue
//                with ONE aggregated atomic per level
// ------------------------------------------------------------------------------------------------

template <typename T> struct MidSub {
    uint32_t start, count, ni, parent;  // start is relative to the item's first position
    T A[6], C[6];
    T cmin, ext;
    uint32_t ax, degen, half, nl;
    uint32_t base[NUM_BUCKETS + 1];     // exclusive bucket offsets inside the sub-node; base[6] = count
    uint32_t child[2];                  // next-level sub-node ids, NONE if the child left the workgroup
    uint32_t heap, _pad;                // heap number of the sub-node (common.hpp heap_child)
    unsigned long long scan0_lo, scanE_lo;  // packed counters before the first / after the last position
    uint32_t scan0_hi, scanE_hi;
};

template <typename T> __device__ __forceinline__ void midsub_derive(MidSub<T>* m) {
    const int ax = largest_axis(m->C);                     // bvh_node.rs:107
    m->ax = (uint32_t)ax;
    m->cmin = m->C[ax];
    m->ext = m->C[3 + ax] - m->C[ax];                      // :108
    m->degen = (m->ext < Traits<T>::eps()) ? 1u : 0u;      // :114
    m->half = m->count / 2;                                // :117
}

struct Packed { unsigned long long lo; uint32_t hi; };  // six 16-bit counters: buckets 0..3 | 4..5
__device__ __forceinline__ Packed packed_onehot(int b) {
    Packed p;
    p.lo = b < 4 ? (1ull << (16 * b)) : 0ull;
    p.hi = (b >= 4 && b < 6) ? (1u << (16 * (b - 4))) : 0u;
    return p;
}
__device__ __forceinline__ uint32_t packed_field(unsigned long long lo, uint32_t hi, int b) {
    return b < 4 ? (uint32_t)((lo >> (16 * b)) & 0xFFFFull) : ((hi >> (16 * (b - 4))) & 0xFFFFu);
}

#ifdef BVH_PROFILE_MID
__device__ unsigned long long g_mid_prof[8];
#define MID_T0() long long _t0 = clock64()
#define MID_T(i) do { long long _t1 = clock64(); if (tid == 0 && blockIdx.x == 0 && true) { atomicAdd(&g_mid_prof[i], (unsigned long long)(_t1 - _t0)); if (i == 5) atomicAdd(&g_mid_prof[0], 1ull); } _t0 = _t1; } while (0)
#else
#define MID_T0()
#define MID_T(i)
#endif

template <typename T, typename Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void k_mid(BuildArgs<T> a, uint32_t first) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    constexpr int MAXN = Cfg::MAXN, MID_THREADS = Cfg::THREADS, PPT = MAXN / MID_THREADS;
    constexpr int HANDOFF = Cfg::HANDOFF;                 // children with at most this many shapes leave the workgroup
    constexpr int MAXSUB = MAXN / (HANDOFF + 1) + 1;      // simultaneously active sub-nodes
    static_assert(MAXSUB <= WAVE, "phase 4 gives one lane of wave 0 to every active sub-node");
    constexpr uint8_t SEG_NONE = 0xFFu;
    __shared__ __attribute__((aligned(16))) T s_box[MAXN * 6];
    __shared__ uint32_t s_idx[MAXN];
    __shared__ uint8_t s_seg[MAXN];
    __shared__ MidSub<T> s_sub[2][MAXSUB];
    __shared__ Key s_keys[MAXSUB * NUM_BUCKETS * STAT_KEYS];
    __shared__ LevelSel<T> s_sel[MAXSUB];                     // phase 4a's outcome per sub-node
    __shared__ uint32_t s_cin[MAXSUB][NUM_BUCKETS];
    __shared__ Key s_sahscr[MID_THREADS / WAVE][72];
    __shared__ unsigned long long s_wlo[MID_THREADS / WAVE];
    __shared__ uint32_t s_whi[MID_THREADS / WAVE];
    __shared__ uint32_t s_nsub;

    const uint32_t n_mid = a.ctr[CTR_MID2];
    const Item<T>* queue = a.mid2;
    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const unsigned long long lt = lanemask_lt();

    for (uint32_t item_id = first + blockIdx.x; item_id < n_mid; item_id += gridDim.x) {
        const Item<T>* it = &queue[item_id];
        const uint32_t istart = it->start, count = it->count;
        const uint32_t* gsrc = a.idx[it->parity];
        uint32_t* gdst = a.idx[it->parity ^ 1];   // where the <= 64-shape children's slices are left
        const uint32_t out_parity = it->parity ^ 1;
        lds_barrier();  // previous item fully done with LDS
        {   // all index loads of the thread, then all its gathers, then the LDS stores (one dependent pair at a time: 12 M shapes 5.90 -> 5.85 ms, 120 k 0.197 -> 0.195)
            uint32_t shq[PPT];
            T bq[PPT][6];
#pragma unroll
            for (int j = 0; j < PPT; j++) { const uint32_t p = tid + (uint32_t)j * MID_THREADS; shq[j] = p < count ? gsrc[istart + p] : 0u; }
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                const T* b = a.aabbs + 6 * (size_t)shq[j];
#pragma unroll
                for (int k = 0; k < 6; k++) bq[j][k] = b[k];
            }
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                const uint32_t p = tid + (uint32_t)j * MID_THREADS;
                if (p < count) {
                    s_idx[p] = shq[j];
                    s_seg[p] = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) s_box[6 * p + k] = bq[j][k];
                } else {
                    s_seg[p] = SEG_NONE;
                }
            }
        }
        if (tid == 0) {
            MidSub<T>* m = &s_sub[0][0];
            m->start = 0; m->count = count; m->ni = it->ni; m->parent = it->parent; m->heap = it->heap;
            for (int k = 0; k < 6; k++) { m->A[k] = it->A[k]; m->C[k] = it->C[k]; }
            midsub_derive(m);
            s_nsub = 1;
        }
        int cur = 0;
        lds_barrier();
        uint32_t nsub = 1;
        while (nsub) {
            MID_T0();
            // ---- reset statistics (consumed in phase 3, two barriers away)
            for (uint32_t j = tid; j < nsub * NUM_BUCKETS * STAT_KEYS; j += MID_THREADS)
                s_keys[j] = key_is_min(j % STAT_KEYS) ? Tr::KEY_POS_INF : Tr::KEY_NEG_INF;
            // ---- phase 1: bucket id of every owned position (bvh_node.rs:204-217)
            T bx[PPT][6];
            uint32_t sid[PPT];
            int sg[PPT], bk[PPT];
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                const uint32_t p = tid * PPT + j;
                sg[j] = (int)s_seg[p];
                bk[j] = 7;
                sid[j] = 0;
                if (sg[j] != SEG_NONE) {
                    const MidSub<T>* m = &s_sub[cur][sg[j]];
#pragma unroll
                    for (int k = 0; k < 6; k++) bx[j][k] = s_box[6 * p + k];
                    sid[j] = s_idx[p];
                    const uint32_t ax = m->ax;
                    const T mn = ax == 0 ? bx[j][0] : (ax == 1 ? bx[j][1] : bx[j][2]);
                    const T mx = ax == 0 ? bx[j][3] : (ax == 1 ? bx[j][4] : bx[j][5]);
                    if (m->degen) bk[j] = (p - m->start) < m->half ? 0 : 1;          // :117
                    else bk[j] = bucket_of(center1(mn, mx), m->cmin, m->ext);        // :210-217
                }
            }
            // ---- phase 2: stable bucket-major sort inside every sub-node (:250-272)
            Packed mine; mine.lo = 0; mine.hi = 0;
#pragma unroll
            for (int j = 0; j < PPT; j++) { Packed o = packed_onehot(bk[j]); mine.lo += o.lo; mine.hi += o.hi; }
            unsigned long long ilo = mine.lo; uint32_t ihi = mine.hi;   // inclusive wave scan
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                unsigned long long ul = __shfl_up(ilo, d);
                uint32_t uh = __shfl_up(ihi, d);
                if (lane >= d) { ilo += ul; ihi += uh; }
            }
            if (lane == WAVE - 1) { s_wlo[wv] = ilo; s_whi[wv] = ihi; }
            lds_barrier();
            MID_T(1);
            unsigned long long rlo = ilo - mine.lo; uint32_t rhi = ihi - mine.hi;
            for (int w2 = 0; w2 < wv; w2++) { rlo += s_wlo[w2]; rhi += s_whi[w2]; }
            unsigned long long plo[PPT]; uint32_t phi[PPT];
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                const uint32_t p = tid * PPT + j;
                plo[j] = rlo; phi[j] = rhi;
                if (sg[j] != SEG_NONE) {
                    MidSub<T>* m = &s_sub[cur][sg[j]];
                    if (p == m->start) { m->scan0_lo = rlo; m->scan0_hi = rhi; }
                    Packed o = packed_onehot(bk[j]); rlo += o.lo; rhi += o.hi;
                    if (p == m->start + m->count - 1) { m->scanE_lo = rlo; m->scanE_hi = rhi; }
                }
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                if (sg[j] == SEG_NONE) continue;
                const uint32_t p = tid * PPT + j;
                MidSub<T>* m = &s_sub[cur][sg[j]];
                const unsigned long long l0 = m->scan0_lo, lE = m->scanE_lo;
                const uint32_t h0 = m->scan0_hi, hE = m->scanE_hi;
                uint32_t acc = 0, mybase = 0;
                uint32_t basev[NUM_BUCKETS + 1];
#pragma unroll
                for (int b = 0; b < NUM_BUCKETS; b++) {
                    basev[b] = acc;
                    if (b == bk[j]) mybase = acc;
                    acc += packed_field(lE, hE, b) - packed_field(l0, h0, b);
                }
                basev[NUM_BUCKETS] = acc;
                if (p == m->start) {
#pragma unroll
                    for (int b = 0; b <= NUM_BUCKETS; b++) m->base[b] = basev[b];
                }
                const uint32_t rank = packed_field(plo[j], phi[j], bk[j]) - packed_field(l0, h0, bk[j]);
                const uint32_t dest = m->start + mybase + rank;
#pragma unroll
                for (int k = 0; k < 6; k++) s_box[6 * dest + k] = bx[j][k];
                s_idx[dest] = sid[j];
            }
            lds_barrier();
            MID_T(2);
            // ---- phase 3: per-(sub-node, bucket) statistics over the sorted order (utils.rs:81-85).
            //      Joins run on floats (one v_min/v_max each, common.hpp join_min/join_max); only a finished run
            //      is converted to integer keys, for the LDS atomics that merge runs across threads and waves.
            {
                T curv[STAT_KEYS], firstv[STAT_KEYS];
                int cur_key = -1, first_key = -1;
                bool have_first = false;
#pragma unroll
                for (int k = 0; k < STAT_KEYS; k++) { curv[k] = key_is_min(k) ? Tr::inf() : -Tr::inf(); firstv[k] = curv[k]; }
                auto flush = [&](int key, const T* v) {   // a complete (or wave-partial) run → LDS statistics
                    Key* kk = s_keys + ((key >> 3) * NUM_BUCKETS + (key & 7)) * STAT_KEYS;
#pragma unroll
                    for (int k = 0; k < STAT_KEYS; k++) {
                        if (key_is_min(k)) atomicMin(&kk[k], Tr::key(v[k]));
                        else atomicMax(&kk[k], Tr::key(v[k]));
                    }
                };
#pragma unroll
                for (int j = 0; j < PPT; j++) {
                    if (sg[j] == SEG_NONE) continue;
                    const uint32_t p = tid * PPT + j;
                    const MidSub<T>* m = &s_sub[cur][sg[j]];
                    const uint32_t rel = p - m->start;
                    int b = 0;
#pragma unroll
                    for (int k = 1; k < NUM_BUCKETS; k++) b += (rel >= m->base[k]) ? 1 : 0;
                    const int key = sg[j] * 8 + b;
                    T v[STAT_KEYS];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const T mn = s_box[6 * p + k], mx = s_box[6 * p + 3 + k];
                        v[k] = mn; v[3 + k] = mx;
                        v[6 + k] = center1(mn, mx); v[9 + k] = v[6 + k];
                    }
                    if (key != cur_key) {
                        if (cur_key >= 0) {
                            if (!have_first) {
                                have_first = true; first_key = cur_key;
#pragma unroll
                                for (int k = 0; k < STAT_KEYS; k++) firstv[k] = curv[k];
                            } else {  // a run that starts and ends inside this thread
                                flush(cur_key, curv);
                            }
                        }
                        cur_key = key;
#pragma unroll
                        for (int k = 0; k < STAT_KEYS; k++) curv[k] = v[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < STAT_KEYS; k++) curv[k] = key_is_min(k) ? join_min(curv[k], v[k]) : join_max(curv[k], v[k]);
                    }
                }
                // wave-level segmented inclusive scan over the threads' LAST runs.  A lane heads a chain unless
                // its only run continues the previous lane's last run; chain start = highest head at or below
                // the lane.  A lane at the start of its chain fetches from itself (join(x, x) = x), and the 12
                // cross-lane moves of a step are issued together.
                const int fk = have_first ? first_key : cur_key;
                const int prev_last = __shfl_up(cur_key, 1);
                const bool cont_prev = lane > 0 && fk >= 0 && fk == prev_last;
                const bool head = have_first || !cont_prev;
                const unsigned long long heads = __ballot(head) | 1ull;
                const unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
                const int chain0 = 63 - __clzll((long long)below);
                T sv[STAT_KEYS];
#pragma unroll
                for (int k = 0; k < STAT_KEYS; k++) sv[k] = curv[k];
                {
                    const int len = lane - chain0;   // lanes of this lane's chain below it
                    const int span = __any(len >= 8) ? 16 : (__any(len >= 4) ? 8 : (__any(len >= 2) ? 4 : (__any(len >= 1) ? 2 : 1)));
                    seg_prefix_scan<T, STAT_KEYS>(sv, lane, chain0, span);
                }
                // the first run of a multi-run thread ends here: join the carry of the previous lane, flush
                {
                    const bool need = have_first && cont_prev;
                    T c[STAT_KEYS];   // the previous lane's value (lane 0: its own) — wave_shr:1
#pragma unroll
                    for (int k = 0; k < STAT_KEYS; k++) c[k] = dpp_fetch<0x138>(sv[k]);
#pragma unroll
                    for (int k = 0; k < STAT_KEYS; k++) {
                        const T j2 = key_is_min(k) ? join_min(firstv[k], c[k]) : join_max(firstv[k], c[k]);
                        firstv[k] = need ? j2 : firstv[k];
                    }
                    if (have_first) flush(first_key, firstv);
                }
                // the last run is flushed by the last lane it reaches inside this wave
                const int next_cont = __shfl_down((int)cont_prev, 1);
                // (a multi-run next lane that continues our run took our value as its carry and flushed it)
                if (cur_key >= 0 && (lane == WAVE - 1 || !next_cont)) flush(cur_key, sv);
            }
            lds_barrier();
            MID_T(3);
            // ---- phase 4a: the SAH selections, one sub-node per wave at a time, each spread over the wave's lanes (lane s of wave 0
            //      running the serial form for sub-node s took 2.9 µs per level whatever the number of sub-nodes)
            for (uint32_t sn = (uint32_t)wv; sn < nsub; sn += MID_THREADS / WAVE) {
                const MidSub<T>* m = &s_sub[cur][sn];
                if (lane < NUM_BUCKETS) s_cin[sn][lane] = m->base[lane + 1] - m->base[lane];
                T A[6];
#pragma unroll
                for (int k = 0; k < 6; k++) A[k] = m->A[k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                sah_select_wave<T>(s_keys + sn * NUM_BUCKETS * STAT_KEYS, s_cin[sn], A, m->degen != 0, &s_sel[sn], s_sahscr[wv], lane);
            }
            lds_barrier();
            MID_T(6);
            // ---- phase 4b: nodes and children (wave 0; lane s owns sub-node s)
            if (wv == 0) {
                const bool has = (uint32_t)lane < nsub;
                MidSub<T>* m = &s_sub[cur][has ? lane : 0];
                T AL[6], CL[6], AR[6], CR[6];
                uint32_t nl = 1, cl = 0, cr = 0;
                if (has) {
                    const LevelSel<T>* sl = &s_sel[lane];
                    nl = sl->nl;
#pragma unroll
                    for (int k = 0; k < 6; k++) { AL[k] = sl->AL[k]; CL[k] = sl->CL[k]; AR[k] = sl->AR[k]; CR[k] = sl->CR[k]; }
                    if (sl->no_winner) atomicOr(&a.ctr[CTR_FLAGS], BUILD_FLAG_EMPTY_SPLIT);
                    cl = nl; cr = m->count - nl;
                    const uint32_t ni = m->ni, li = ni + 1, ri = li + (2 * nl - 1);  // bvh_node.rs:138-142
                    typename Tr::Node* nd = &a.nodes[ni];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        nd->l_min[k] = AL[k]; nd->l_max[k] = AL[3 + k];
                        nd->r_min[k] = AR[k]; nd->r_max[k] = AR[3 + k];
                    }
                    nd->parent = m->parent; nd->l = li; nd->r = ri; nd->shape = NONE;
                    a.node_start[ni] = istart + m->start;
                    a.node_count[ni] = m->count;
                    a.node_slot[ni] = (uint16_t)m->heap;
                    m->nl = nl;
                }
                const bool subL = has && cl > (uint32_t)HANDOFF, subR = has && cr > (uint32_t)HANDOFF;
                // a child that leaves goes to the wave tier (<= 64 shapes); larger ones stay in this workgroup
                const bool smL = has && !subL && cl <= (uint32_t)SMALL_MAX, smR = has && !subR && cr <= (uint32_t)SMALL_MAX;
                const bool m2L = has && !subL && !smL, m2R = has && !subR && !smR;
                const unsigned long long mL = __ballot(subL), mR = __ballot(subR);
                const unsigned long long qL = __ballot(smL), qR = __ballot(smR);
                const unsigned long long wL = __ballot(m2L), wR = __ballot(m2R);
                const uint32_t idL = (uint32_t)(__popcll(mL & lt) + __popcll(mR & lt));
                const uint32_t idR = idL + (subL ? 1u : 0u);
                const uint32_t nsmall = (uint32_t)(__popcll(qL) + __popcll(qR));
                const uint32_t nmid2 = (uint32_t)(__popcll(wL) + __popcll(wR));
                uint32_t sbase = 0, wbase = 0;
#ifdef BVH_PROFILE_MID
                const long long _ta = clock64();
#endif
                if (lane == 0 && nsmall) sbase = atomicAdd(&a.ctr[CTR_SMALL], nsmall);
                if (lane == 0 && nmid2) wbase = atomicAdd(&a.ctr[CTR_MID2], nmid2);
                sbase = __shfl(sbase, 0);
                wbase = __shfl(wbase, 0);
#ifdef BVH_PROFILE_MID
                asm volatile("" ::"v"(sbase), "v"(wbase));
                if (tid == 0 && blockIdx.x == 0) atomicAdd(&g_mid_prof[7], (unsigned long long)(clock64() - _ta));   // the queue-slot atomics' round trip
#endif
                const uint32_t slL = sbase + (uint32_t)(__popcll(qL & lt) + __popcll(qR & lt));
                const uint32_t slR = slL + (smL ? 1u : 0u);
                const uint32_t w2L = wbase + (uint32_t)(__popcll(wL & lt) + __popcll(wR & lt));
                const uint32_t w2R = w2L + (m2L ? 1u : 0u);
                if (has) {
                    const uint32_t ni = m->ni, li = ni + 1, ri = li + (2 * nl - 1);
                    m->child[0] = subL ? idL : NONE;
                    m->child[1] = subR ? idR : NONE;
                    for (int side = 0; side < 2; side++) {
                        const bool is_sub = side ? subR : subL;
                        const uint32_t cstart = side ? m->start + nl : m->start;
                        const uint32_t ccount = side ? cr : cl;
                        const uint32_t cni = side ? ri : li;
                        const T* CA = side ? AR : AL;
                        const T* CC = side ? CR : CL;
                        if (is_sub) {
                            MidSub<T>* c = &s_sub[cur ^ 1][side ? idR : idL];
                            c->start = cstart; c->count = ccount; c->ni = cni; c->parent = ni;
                            c->heap = heap_child(m->heap, (uint32_t)side);
#pragma unroll
                            for (int k = 0; k < 6; k++) { c->A[k] = CA[k]; c->C[k] = CC[k]; }
                            midsub_derive(c);
                        } else {
                            const bool to_small = side ? smR : smL;
                            Item<T>* g = to_small ? &a.small[side ? slR : slL] : &a.mid2[side ? w2R : w2L];
                            g->ni = cni; g->parent = ni; g->start = istart + cstart; g->count = ccount;
                            g->tile_base = 0; g->parity = out_parity; g->heap = heap_child(m->heap, (uint32_t)side); g->_r1 = 0;
#pragma unroll
                            for (int k = 0; k < 6; k++) { g->A[k] = CA[k]; g->C[k] = CC[k]; }
                        }
                    }
                }
                if (lane == 0) s_nsub = (uint32_t)(__popcll(mL) + __popcll(mR));
            }
            lds_barrier();
            MID_T(4);
            // ---- phase 5: positions follow their sub-node's child; slices of children that leave the
            //      workgroup (<= 64 shapes) are written to the global index buffer for the wave tier
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                if (sg[j] == SEG_NONE) continue;
                const uint32_t p = tid * PPT + j;
                const MidSub<T>* m = &s_sub[cur][sg[j]];
                const uint32_t ch = m->child[(p - m->start) < m->nl ? 0 : 1];
                if (ch != NONE) s_seg[p] = (uint8_t)ch;
                else { s_seg[p] = SEG_NONE; gdst[istart + p] = s_idx[p]; }
            }
            lds_barrier();
            MID_T(5);
            cur ^= 1;
            nsub = s_nsub;
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

#ifdef BVH_SMALL_PROFILE   // developer build (tools/small_prof.py): 100 MHz wall-clock stamps per wave of k_small
__device__ unsigned long long g_small_prof[20 * 8192];   // [0] entry, [1] item + shapes loaded, [2 + L] level L finished (L < 16), [18] exit, [19] shapes
void debug_small_prof(unsigned long long* out, size_t n) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_prof), sizeof(unsigned long long) * n);
}
#endif
// FLAT_INLINE: the wave also writes its subtree's part of the flatten (a.fl_parts; f32).  A kernel of its own because that part needs 119 registers (4 waves per
// SIMD) where the build alone needs 40: a build that is not followed by its flatten keeps 8 (600 k shapes: 0.414 against 0.429 ms)
template <typename T, bool FLAT_INLINE> __global__ __launch_bounds__(256) void k_small(BuildArgs<T> a, uint32_t first) {
    using Tr = Traits<T>;
#ifdef BVH_SMALL_PROFILE
    const unsigned long long sp_t0 = wall_clock64();
#endif
    const uint32_t n_small = a.ctr[CTR_SMALL];
    const uint32_t wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt();
    __shared__ uint32_t s_inner[FLAT_INLINE ? 256 / WAVE : 1][WAVE];   // per wave: the inner nodes of the subtree it is building, in the order it creates them (the wave's flatten below)
    uint32_t* my_inner = s_inner[FLAT_INLINE ? (threadIdx.x >> 6) : 0];
    for (uint32_t wave = first + wave0; wave < n_small; wave += nwaves) {
    uint32_t ninner = 0;       // (wave-uniform)
    bool wave_empty = false;   // (wave-uniform) a split without a SAH winner: the children's stored boxes are EMPTY, not their shapes' (bvh_node.rs:225-230)
    const Item<T>* it = &a.small[wave];
    const uint32_t istart = it->start;
    const int n = (int)it->count;
    const uint32_t ni0 = it->ni;   // the subtree's nodes: [ni0, ni0 + 2 n - 1), pre-order
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
    uint32_t ni = it->ni, parent = it->parent, heap = it->heap;
    T Cb[6], A0[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { Cb[k] = it->C[k]; A0[k] = it->A[k]; }
    T saA = surface_area(A0);
#ifdef BVH_SMALL_PROFILE
    int sp_level = 0;
    const bool sp_on = wave < 8192 && wave == first + wave0 && lane == 0;
    if (sp_on) {
        g_small_prof[20 * wave] = sp_t0;
        g_small_prof[20 * wave + 1] = wall_clock64() + (unsigned long long)(box[0] != box[0] ? 1 : 0);   // (after the shape loads have landed)
        g_small_prof[20 * wave + 19] = (unsigned long long)n;
        for (int i = 2; i < 19; i++) g_small_prof[20 * wave + i] = 0;
    }
#endif

    while (true) {
#ifdef BVH_SMALL_PROFILE
        if (sp_on && sp_level > 0 && sp_level <= 16) g_small_prof[20 * wave + 1 + sp_level] = wall_clock64();
        sp_level++;
#endif
        int segn = hi - lo;
        if (!done && segn == 1) {  // bvh_node.rs:95-104
            typename Tr::Node* nd = &a.nodes[ni];
#pragma unroll
            for (int k = 0; k < 3; k++) { nd->l_min[k] = 0; nd->l_max[k] = 0; nd->r_min[k] = 0; nd->r_max[k] = 0; }
            nd->parent = parent; nd->l = NONE; nd->r = NONE; nd->shape = shape;
            a.shape_node[shape] = ni;       // set_bh_node_index (:102)
            a.node_slot[ni] = (uint16_t)heap;
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

        // ---- segmented inclusive prefix (P) and suffix (S) joins of AABB and centroid bounds over the lanes of the segment
        //      (seg_prefix_scan / seg_suffix_scan above: DPP inside a row, one-lane carries across rows)
        // (the AABBs only: the children's centroid bounds are reduced over the NEW segments once the split is known — 18
        //  scanned values and 18 fetches per level instead of 24 and 24)
        T P[6], S[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { P[k] = box[k]; S[k] = box[k]; }
        const int span = __any(!done && segn > 8) ? 16 : (__any(!done && segn > 4) ? 8 : (__any(!done && segn > 2) ? 4 : (__any(!done && segn > 1) ? 2 : 1)));
        seg_prefix_scan<T, 6>(P, lane, lo, span);
        seg_suffix_scan<T, 6>(S, lane, hi, span);
        // ---- SAH cost of the 5 candidates (:231-247): L = prefix at the last lane of bucket <= s,
        //      R = suffix at the first lane of bucket > s
        const T saP = surface_area(P), saS = surface_area(S);
        int cumv[NUM_BUCKETS - 1];
        T salv[NUM_BUCKETS - 1], sarv[NUM_BUCKETS - 1];
        {
            int cum = 0;
#pragma unroll
            for (int s2 = 0; s2 < NUM_BUCKETS - 1; s2++) { cum += cnt[s2]; cumv[s2] = cum; }
        }
#pragma unroll
        for (int s2 = 0; s2 < NUM_BUCKETS - 1; s2++) {   // the 10 fetches first, the costs afterwards
            const int q2 = lo + cumv[s2];
            // (no clamps: ds_bpermute takes the lane modulo 64, and a position outside the segment — q2 - 1 < lo or q2 >= hi — only
            //  arises for a side without shapes, which the non-degenerate selection cannot produce: bucket 0 and bucket 5 are never empty)
            salv[s2] = lane_fetch(saP, (q2 - 1) << 2);
            sarv[s2] = lane_fetch(saS, q2 << 2);
        }
        int nl = cnt[0];
        bool taken = false;  // no winner (NaN/inf costs) → min_bucket 0 with EMPTY child bounds (:225-230)
        T min_cost = Tr::inf();
#pragma unroll
        for (int s2 = 0; s2 < NUM_BUCKETS - 1; s2++) {
            T cl = (T)cumv[s2] * salv[s2];
            T cr = (T)(segn - cumv[s2]) * sarv[s2];
            T num = cl + cr;
            T cost = num / saA;
            bool take = degen ? (s2 == 0) : (cost < min_cost);
            if (take) { min_cost = cost; nl = cumv[s2]; taken = true; }
        }
        const int q = lo + nl;
        const int ql4 = (q - 1) << 2, qr4 = q << 2;   // (an empty side only without a winner: its bounds are replaced below)
        T AL[6], AR[6], Cn[6];
        const bool left = lane < q;
        {
#pragma unroll
            for (int k = 0; k < 6; k++) AL[k] = lane_fetch(P[k], ql4);
#pragma unroll
            for (int k = 0; k < 6; k++) AR[k] = lane_fetch(S[k], qr4);
            // centroid bounds of the lane's child = join over its new segment [nlo, nhi): prefix scan + the last lane's value
            const int nlo = done ? lane : (left ? lo : q), nhi = done ? lane + 1 : (left ? q : hi);
            T Cq[6];
#pragma unroll
            for (int k = 0; k < 3; k++) { Cq[k] = c[k]; Cq[3 + k] = c[k]; }
            seg_prefix_scan<T, 6>(Cq, lane, nlo, span);
            const int last4 = (nhi - 1) << 2;
#pragma unroll
            for (int k = 0; k < 6; k++) Cn[k] = lane_fetch(Cq[k], last4);
        }
        if (!taken) { box_empty(AL); box_empty(AR); box_empty(Cn); }
        if (!done && !taken && lane == lo) atomicOr(&a.ctr[CTR_FLAGS], BUILD_FLAG_EMPTY_SPLIT);
        if constexpr (FLAT_INLINE) {   // the nodes this level creates (one per segment, by its first lane), filed for the wave's flatten
            wave_empty = wave_empty || __any(!done && !taken);
            const unsigned long long cm = __ballot(!done && lane == lo);
            if (!done && lane == lo) my_inner[ninner + (uint32_t)__popcll(cm & lt)] = ni;
            ninner += (uint32_t)__popcll(cm);
        }
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
                a.node_slot[ni] = (uint16_t)heap;
            }
            parent = ni;
            if (left) { hi = q; ni = li; saA = surface_area(AL); heap = heap_child(heap, 0u); }
            else { lo = q; ni = ri; saA = surface_area(AR); heap = heap_child(heap, 1u); }
#pragma unroll
            for (int k = 0; k < 6; k++) Cb[k] = Cn[k];
        }
    }
    // ---- the subtree is complete: its part of the flatten, by the wave that still has it in its caches (BVHGPU_TUNE_FLATTEN_INLINE).  FLAT
    //      and WIDE only read records of the subtree itself and of the item's parent (written by the launch that queued the item); TRAV
    //      reads a node BEHIND the subtree and stays with k_flatten.
    if constexpr (FLAT_INLINE && sizeof(T) == 4) if (a.fl_parts != 0u) {   // (uniform; f32 only: an f64 wide node is 64 registers)
        // this wave's own stores (node records, start / count / slot) are read back below: workgroup scope is enough (one CU, one L1) — a device-scope
        // fence writes the XCD's L2 back (0.1 ms per item on this chip)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const bool with_flat = (a.fl_parts & (uint32_t)FLATTEN_FLAT) != 0u;
        const bool fast = n >= 2 && !(with_flat && wave_empty);   // (wave-uniform)
        if (fast) {
            // ONE round: lane j takes the j-th inner node the wave created (at most 63) through the shared body; a leaf has no wide node, and its two
            // FlatNode entries need no load at all — the lane still holds its shape, its node index and its box, and the box its parent stored for it is
            // that box bit for bit (the prefix / suffix join over a segment of one lane is the lane's own box; a split without a winner stores EMPTY boxes
            // instead: such a subtree takes the general loop below).  Leaves first: what the loop kept per lane dies before the inner nodes' registers are needed
            if (with_flat && lane < n) {
                const uint32_t nav = ni - 1u + (istart + (uint32_t)lane);   // flatten_node.hpp: nav(i) = i - 1 + L_i
                typename Tr::Flat fe = {}, lf = {};
#pragma unroll
                for (int k = 0; k < 3; k++) { fe.min[k] = box[k]; fe.max[k] = box[3 + k]; lf.min[k] = Tr::inf(); lf.max[k] = -Tr::inf(); }
                fe.entry = nav + 1u; fe.exit = nav + 2u; fe.shape = NONE;    // exit = nav + 3 k - 1, k = 1
                lf.entry = NONE; lf.exit = nav + 2u; lf.shape = shape;       // flat_bvh.rs:129-141
                a.fl.flat[nav] = fe;
                a.fl.flat[nav + 1u] = lf;
            }
        }
        // the nodes that go through the shared body: the inner nodes from the wave's list (one round), or — the general case — every node of the subtree
        const uint32_t nn_item = 2u * (uint32_t)n - 1u;
        const uint32_t todo = fast ? ninner : nn_item;
#pragma unroll 1
        for (uint32_t j = (uint32_t)lane; j < todo; j += WAVE) {
            const uint32_t i = fast ? my_inner[j] : ni0 + j;
            const typename Tr::Node ndf = a.fl.nodes[i];
            if (with_flat) flatten_node<T, FLATTEN_FLAT | FLATTEN_WIDE>(a.fl, i, ndf);
            else flatten_node<T, FLATTEN_WIDE>(a.fl, i, ndf);
        }
    }
#ifdef BVH_SMALL_PROFILE
    if (sp_on) g_small_prof[20 * wave + 18] = wall_clock64();
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// host driver.  build_enqueue puts the whole optimistic schedule (and the counter readback) on the stream without
// a host round trip; build_finalize waits for it, validates the input contract and finishes an unbalanced tree level by
// level.  The synchronous entry points run one after the other; the *_async entry points of the C ABI return after the
// enqueue and leave the finalize to bvhgpu_tree_wait / bvhgpu_hits_wait.
// ------------------------------------------------------------------------------------------------
// sizes and offsets of the rotating arrays of the one-launch-per-level tier inside t->lvbuf
template <typename T> struct LevelLayout {
    size_t n_slots, n_tiles, sz_tile_item, sz_tile_cnt, sz_stats, off_tile_item, off_tile_cnt, off_stats, bytes;
    LevelLayout(size_t n, size_t mid_max) {
        n_slots = n / (mid_max + 1) + 2;
        n_tiles = n / TILE + n_slots + 4;
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        sz_tile_item = up(n_tiles * 16);
        sz_tile_cnt = up(n_tiles * NUM_BUCKETS * 4);
        sz_stats = up(n_slots * STAT_REP * sizeof(ItemStats<T>));
        off_tile_item = 0;
        off_tile_cnt = off_tile_item + 3 * sz_tile_item;
        off_stats = off_tile_cnt + 3 * sz_tile_cnt;
        bytes = off_stats + 3 * sz_stats;
    }
};

template <typename T> static bool level_fused(const bvhgpu_tree* t) {
    const int v = t->ctx->tune[BVHGPU_TUNE_BUILD_LEVEL_LAUNCHES];
    return v == 1 || (v != 2 && t->n <= MID_SCENE_SPLIT);
}
// Positions per tile of the two-launch schedule.  Every tile is one pass of a workgroup through a chain of dependent fetches (tile -> item -> counts ->
// offsets) ahead of its shapes, so a level of millions of positions is cheaper in fewer, longer tiles — and a level of 360 k in many short ones
// (build ms by tile, tools/level_tile_sweep.py, profiles/r6_level_tile_sweep.log: 12 M shapes 6.99 / 6.38 / 6.19 / 6.12 at 512 / 1024 / 2048 /
// 4096; 3.6 M 2.15 / 1.99 / 1.96 / 2.01; 1.2 M 0.83 / 0.77 / 0.77 / 0.87; 360 k 0.35 / 0.35 / 0.40 / 0.50).  A scheduling unit only: the trees
// are byte-equal whatever the tile (same sweep).  The fused schedule's kernels are written for TILE.
template <typename T> static uint32_t level_tile(const bvhgpu_tree* t) {
    if (level_fused<T>(t)) return (uint32_t)TILE;
    const int v = t->ctx->tune[BVHGPU_TUNE_BUILD_LEVEL_TILE];
    if (v > 0) return (uint32_t)std::min(std::max((v + 255) / 256 * 256, (int)TILE), 16384);   // (the buffers are sized for TILE-position tiles: never fewer positions)
    return t->n >= 6000000 ? 4096u : t->n >= 2000000 ? 2048u : t->n >= 600000 ? 1024u : (uint32_t)TILE;
}

template <typename T> static BuildArgs<T> make_args(bvhgpu_tree* t, const T* src) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    BuildArgs<T> a;
    a.aabbs = t->aabbs.as<T>();
    a.src = src ? src : t->aabbs.as<T>();
    a.nodes = t->nodes.as<typename Tr::Node>();
    a.node_start = t->node_start.as<uint32_t>();
    a.node_count = t->node_count.as<uint32_t>();
    a.shape_node = t->shape_node.as<uint32_t>();
    a.node_slot = t->node_slot.as<uint16_t>();
    a.slot_entry = t->slot_entry.as<uint32_t>();
    a.n_slots = TopCfg<T>::SLOTS;
    a.wslot_node = t->wslot_node.as<uint32_t>();
    a.idx[0] = t->idx[0].as<uint32_t>(); a.idx[1] = t->idx[1].as<uint32_t>();
    a.bk = t->bk.as<uint8_t>();
    a.big[0] = t->big[0].as<Item<T>>(); a.big[1] = t->big[1].as<Item<T>>();
    a.mid2 = t->mid2.as<Item<T>>();
    a.small = t->small.as<Item<T>>();
    a.stats[0] = t->stats[0].as<ItemStats<T>>(); a.stats[1] = t->stats[1].as<ItemStats<T>>();
    a.tile_item[0] = t->tile_item[0].as<uint32_t>(); a.tile_item[1] = t->tile_item[1].as<uint32_t>();
    a.tile_cnt = t->tile_cnt.as<uint32_t>();
    a.tile2 = level_tile<T>(t);
    a.fl = FlattenArgs<T>{}; a.fl_parts = 0;   // (build_enqueue sets them when the flatten behind the build is known)
    a.n_chunks = (uint32_t)(t->chunk_cnt.cap / (2 * 2 * NUM_BUCKETS * 4));
    a.chunk_cnt[0] = t->chunk_cnt.as<uint32_t>();
    a.chunk_cnt[1] = a.chunk_cnt[0] ? a.chunk_cnt[0] + (size_t)a.n_chunks * 2 * NUM_BUCKETS : nullptr;
    a.ctr = t->ctr.as<uint32_t>();
    a.rootkeys = reinterpret_cast<Key*>(reinterpret_cast<char*>(t->ctr.p) + ROOTKEY_OFF);
    a.n = (uint32_t)t->n;
    a.xbar = t->xbar.as<unsigned long long>();   // (NULL unless the persistent level tier is in use)
    a.prep_wgs = 0;
    const bool small_scene = t->n <= MID_SCENE_SPLIT;
    a.mid_max = (uint32_t)(small_scene ? MidSmallScene<T>::MAXN : MidLargeScene<T>::MAXN);
    // level tier with one launch per level (LevelArgs): carved out of t->lvbuf by level_layout()
    const LevelLayout<T> ly(t->n, a.mid_max);
    a.lv.slot_div = a.mid_max + 1u; a.lv.n_slots = (uint32_t)ly.n_slots; a.lv.n_tiles = (uint32_t)ly.n_tiles;
    char* base = t->lvbuf.cap >= ly.bytes ? reinterpret_cast<char*>(t->lvbuf.p) : nullptr;   // (only sized when that schedule is used)
    for (int i = 0; i < 3; i++) {
        a.lv.tile_map[i] = base ? reinterpret_cast<uint4*>(base + ly.off_tile_item + i * ly.sz_tile_item) : nullptr;
        a.lv.tile_cnt[i] = base ? reinterpret_cast<uint32_t*>(base + ly.off_tile_cnt + i * ly.sz_tile_cnt) : nullptr;
        a.lv.stats[i] = base ? reinterpret_cast<ItemStats<T>*>(base + ly.off_stats + i * ly.sz_stats) : nullptr;
    }
    a.lv.item[0] = a.big[0]; a.lv.item[1] = a.big[1];
    a.lv.bk[0] = a.bk; a.lv.bk[1] = a.bk ? a.bk + t->n : nullptr;
    return a;
}

template <typename T> struct BuildGrid {
    size_t max_big, max_mid2, max_tiles;
    int tile_grid, sel_grid, mid2_grid, small_grid, level_grid;
    BuildGrid(const bvhgpu_tree* t, size_t mid_max) {
        const size_t n = t->n;
        max_big = n / (mid_max + 1) + 2;       // simultaneously active nodes with > mid_max shapes
        max_mid2 = n / (SMALL_MAX + 1) + 2;    // workgroup tier: nodes with 65..mid_max shapes
        max_tiles = n / TILE + max_big + 2;
        tile_grid = (int)std::min<size_t>(n / level_tile<T>(t) + max_big + 2, 2048);
        sel_grid = (int)std::min<size_t>((max_big + 3) / 4, 1024);
        mid2_grid = (int)std::min<size_t>(max_mid2, (size_t)t->ctx->n_cu * 4);
        small_grid = (int)std::min<size_t>((n + 3) / 4, (size_t)t->ctx->n_cu * 8);
        level_grid = (int)std::min<size_t>(LevelLayout<T>(n, mid_max).n_tiles, 2048);
    }
};

// One level-synchronous pass: after pass L (counted from 0) the items of level L are split and their children queued.
//   two launches per level (scenes above MID_SCENE_SPLIT shapes; BVHGPU_TUNE_BUILD_LEVEL_LAUNCHES = 2): k_bin(L), k_split(L);
//   one launch per level (smaller scenes; = 1): the root is binned by k_level<ROOT> ahead of pass 0, pass L = k_level(L + 1), which splits
//   level L and bins level L + 1 in the same launch.
// Which schedule: the fused launch shortens the dependent chain (what small scenes are bound by: 0.200 against 0.220 ms at
// 120 k shapes) and does more work per shape (selection per tile, three rotating accumulator sets, binning inside the rewrite:
// 0.374 against 0.349 ms at 360 k, 10.5 against 8.4 ms at 12 M) — so it is used up to MID_SCENE_SPLIT shapes.
template <typename T> static void run_level(bvhgpu_tree* t, const BuildArgs<T>& a, const BuildGrid<T>& g, int L) {
    hipStream_t st = t->ctx->stream;
    if (level_fused<T>(t)) {
        if (L == 0) hipLaunchKernelGGL((k_level<T, true>), dim3(g.level_grid), dim3(LEVEL_THREADS), 0, st, a, 0);
        hipLaunchKernelGGL((k_level<T, false>), dim3(g.level_grid), dim3(LEVEL_THREADS), 0, st, a, L + 1);
        return;
    }
    hipLaunchKernelGGL(k_bin<T>, dim3(g.tile_grid), dim3(256), 0, st, a, L);
    hipLaunchKernelGGL(k_split<T>, dim3(g.sel_grid + g.tile_grid), dim3(256), 0, st, a, L, (uint32_t)g.sel_grid);
}
// workgroup tier over the items queued from `mid2_done` on, then the wave tier from `small_done` on
template <typename T> static void run_lower_tiers(bvhgpu_tree* t, const BuildArgs<T>& a, const BuildGrid<T>& g, uint32_t mid2_done,
                                                  uint32_t small_done) {
    hipStream_t st = t->ctx->stream;
    if (t->n > (size_t)SMALL_MAX) {
        if (t->n <= MID_SCENE_SPLIT) hipLaunchKernelGGL((k_mid<T, MidSmallScene<T>>), dim3(g.mid2_grid), dim3(MidSmallScene<T>::THREADS), 0, st, a, mid2_done);
        else hipLaunchKernelGGL((k_mid<T, MidLargeScene<T>>), dim3(g.mid2_grid), dim3(MidLargeScene<T>::THREADS), 0, st, a, mid2_done);
    }
    if (a.fl_parts != 0u) hipLaunchKernelGGL((k_small<T, true>), dim3(g.small_grid), dim3(256), 0, st, a, small_done);
    else hipLaunchKernelGGL((k_small<T, false>), dim3(g.small_grid), dim3(256), 0, st, a, small_done);
}

// redo: build_finalize builds the SAME generation again (the persistent level tier gave up): no new generation, the tree's own AABB copy
template <typename T> void build_enqueue(bvhgpu_tree* t, const T* aabbs_dev, size_t n, bool flatten_after, bool redo) {
    using Tr = Traits<T>;
    using Key = typename Tr::Key;
    bvhgpu_ctx* ctx = t->ctx;
    hipStream_t st = ctx->stream;
    join_flat(t);   // (a flatten part still running on the side stream reads the arrays this build is about to overwrite)
    t->built = false; t->flattened = false; t->lazy_flat = false; t->pending_build = false; t->exact_only = false; t->pending_recv = false;
    t->pend_persist = false;
    if (!redo) t->gen++;   // results enqueued from here on belong to this build (bvhgpu_hits_wait compares generations)
    if (n != t->n) t->has_tris = false;   // one triangle per shape: a different shape count invalidates the vertex array
    t->n = n; t->n_nodes = n ? 2 * n - 1 : 0;
    t->n_flat = n >= 2 ? 3 * n - 2 : n;
    t->n_trav = n >= 2 ? 2 * n - 2 : n;
    t->levels = 0;
    if (n == 0) { t->built = true; t->flattened = flatten_after; return; }

    const size_t MID_MAX = n <= MID_SCENE_SPLIT ? (size_t)MidSmallScene<T>::MAXN : (size_t)MidLargeScene<T>::MAXN;
    const BuildGrid<T> g(t, MID_MAX);
    t->aabbs.reserve(n * 6 * sizeof(T));
    // the optimistic flatten may run over nodes an unfinished build has not written yet: never-written memory must at least
    // be harmless (k_flatten also bounds-checks what it reads from a node)
    if (t->nodes.reserve(t->n_nodes * sizeof(typename Tr::Node))) BVH_HIP(hipMemsetAsync(t->nodes.p, 0, t->nodes.cap, st));
    if (t->node_start.reserve(t->n_nodes * 4)) BVH_HIP(hipMemsetAsync(t->node_start.p, 0, t->node_start.cap, st));
    if (t->node_count.reserve(t->n_nodes * 4)) BVH_HIP(hipMemsetAsync(t->node_count.p, 0, t->node_count.cap, st));
    t->shape_node.reserve(n * 4);
    if (t->node_slot.reserve(t->n_nodes * 2)) BVH_HIP(hipMemsetAsync(t->node_slot.p, 0xFF, t->node_slot.cap, st));
    t->slot_entry.reserve(TopCfg<T>::SLOTS * 4);
    t->wslot_node.reserve(WIDE_SLOTS * 4);
    t->idx[0].reserve(n * 4);
    t->idx[1].reserve(n * 4);
    t->bk.reserve(2 * n);   // (the one-launch-per-level tier keeps the buckets of two consecutive levels)
    if (level_fused<T>(t) && n > (size_t)MID_MAX) t->lvbuf.reserve(LevelLayout<T>(n, MID_MAX).bytes);
    // Persistent level tier (k_level<T, false, true>): where the eight level-3 subtrees are big enough to be worth a workgroup group each.  A tree on
    // which it once gave up (its workgroups were not resident together: another stream kept the CUs) stays with a launch per level.
    const int persist_knob = ctx->tune[BVHGPU_TUNE_BUILD_LEVEL_PERSIST];
    // (a tree on which the persistent tier once gave up tries it again 64 builds later: whatever kept its workgroups apart may be gone)
    if (t->persist_broken && persist_knob != 0 && !redo && t->persist_retry_in > 0 && --t->persist_retry_in == 0) t->persist_broken = false;
    const bool persist = persist_knob != 0 && !t->persist_broken && level_fused<T>(t) && n >= 32 * (MID_MAX + 1);
    if (persist) t->xbar.reserve(XBAR_WORDS * sizeof(unsigned long long));
    for (int i = 0; i < 2; i++) {
        t->big[i].reserve(g.max_big * sizeof(Item<T>));
        t->stats[i].reserve(g.max_big * STAT_REP * sizeof(ItemStats<T>));
        t->tile_item[i].reserve(g.max_tiles * 4);
    }
    t->mid2.reserve(g.max_mid2 * sizeof(Item<T>));
    t->small.reserve((n + 1) * sizeof(Item<T>));
    t->tile_cnt.reserve(g.max_tiles * NUM_BUCKETS * 4);
    if (!level_fused<T>(t)) t->chunk_cnt.reserve((g.max_tiles / CHUNK_TILES + 2) * 2 * 2 * NUM_BUCKETS * 4);   // (two parities x two slots per block)
    if (t->ctr.reserve(ROOTKEY_OFF + (size_t)PREP_MAX_WG * STAT_KEYS * sizeof(Key))) t->ctr_ready = false;   // a fresh buffer has not been zeroed by the previous build
    if (!t->pin) BVH_HIP(hipHostMalloc(&t->pin, ROOTKEY_OFF, hipHostMallocDefault));

    BuildArgs<T> a = make_args<T>(t, aabbs_dev);
    // counters and root keys were reset at the end of the previous build of this tree (off the critical path)
    if (!t->ctr_ready) hipLaunchKernelGGL(k_init<T>, dim3(1), dim3(256), 0, st, a);
    t->ctr_ready = false;
#ifndef BVH_PREP_PER_WG
#define BVH_PREP_PER_WG 1024
#endif
    const int prep_grid = (int)std::min<size_t>((n + BVH_PREP_PER_WG - 1) / BVH_PREP_PER_WG, (size_t)PREP_MAX_WG);   // 256 / 512 / 1024 / 2048 shapes per workgroup measured
    a.prep_wgs = (uint32_t)prep_grid;
    hipLaunchKernelGGL(k_prep<T>, dim3(prep_grid), dim3(256), 0, st, a);   // + aabbs copy + input validation
    // the root item: by the level tier's first launch itself (one launch per level), else by a one-workgroup launch
    if (!(level_fused<T>(t) && n > (size_t)MID_MAX)) hipLaunchKernelGGL(k_root<T>, dim3(1), dim3(256), 0, st, a, (uint32_t)prep_grid, 0);

    // Optimistic schedule with no host round trip: enough level-synchronous passes for a balanced
    // tree, then the workgroup tier over everything queued so far, then the wave tier.  ONE readback
    // at the end checks that nothing is left in the level queue; unbalanced trees continue from there.
    int level = 0;
    if (n > (size_t)MID_MAX) {
        // How many level-synchronous passes to enqueue blind: for a first build the levels of a balanced tree plus one (an unbalanced
        // tree continues in build_finalize, one host round trip per level); for a REbuild of as many shapes what the previous
        // build of this tree needed — a frame loop's scene changes little from one build to the next, so neither a wasted
        // empty pass (~4.5 µs) nor, for an unbalanced scene, the slow path with its replay of the batch is paid every frame.
        int fixed = 1;
        for (size_t m = n; m > (size_t)MID_MAX; m = (m + 1) / 2) fixed++;
        if (t->hint_levels > 0 && t->hint_n == n) fixed = t->hint_levels;
        if (fixed > MAXLV - 4) fixed = MAXLV - 4;
        if (persist) {
            // tree levels 0 .. 2 are split (and level 3 binned) by launches over the whole chip; everything below by ONE persistent launch,
            // a workgroup group per level-3 subtree, however deep the subtrees turn out to be (no blind pass count, no host loop)
            for (; level < 3; level++) run_level<T>(t, a, g, level);
            const int nx = persist_knob >= 8 ? std::min(persist_knob, 64) : 32;
            hipLaunchKernelGGL((k_level<T, false, true>), dim3(8 * nx), dim3(LEVEL_THREADS), 0, st, a, 4);
            t->pend_persist = true;
            fixed = level;
        }
        for (; level < fixed; level++) {
            run_level<T>(t, a, g, level);
            if (level == 3 && ctx->tune[BVHGPU_TUNE_WIDE_EARLY_ITEMS] != 0 && !t->pend_persist) {   // (only when the early item filter is switched on: an event record costs the level chain a gap)
                // tree levels 0..3 are split: their BvhNode records (and with them the boxes of the 16 subtrees the wide
                                // walk cuts its rays into) are final — a batch enqueued behind this build may filter its rays from here on
                if (!t->ev_top) BVH_HIP(hipEventCreateWithFlags(&t->ev_top, hipEventDisableTiming));
                BVH_HIP(hipEventRecord(t->ev_top, st));
                t->ev_top_gen = t->gen;
            }
        }
    }
    // the flatten behind this build: its FLAT / WIDE parts for every node of at most SMALL_MAX shapes by the wave tier itself
    uint32_t inline_parts = 0;
    // (measured, tools/gpu_flatten_inline.sh: 12 M shapes 16.15 -> 15.31 ms per step; 120 k shapes, 6 + 3 alternating pairs of processes: 2 973 -> 3 006
    //  Mrays/s on average — the flatten kernel is the step's HBM-bound launch, inside the wave tier's dependent chain its stores cost next to nothing)
    if (flatten_after && n >= 2 && sizeof(T) == 4 && ctx->tune[BVHGPU_TUNE_FLATTEN_INLINE] != 0) {   // (f64: a wide node is 64 registers — k_flatten keeps it)
        const FlattenPlan p = flatten_plan<T>(t, ctx->tune[BVHGPU_TUNE_FLATTEN_LAZY] != 0);
        if (p.with_wide) {
            inline_parts = (uint32_t)(p.parts & (FLATTEN_FLAT | FLATTEN_WIDE));
            a.fl = flatten_args<T>(t, p.with_wide, p.with_guide); a.fl_parts = inline_parts;
        }
    }
    run_lower_tiers<T>(t, a, g, 0u, 0u);
    // counters: readback through the pinned page + reset for the next build, by the flatten kernel when there is one
    // (optimistic too: redone if the build turns out to be unfinished) and by a launch of their own otherwise
    // (the same block leaves a status word in HBM — build flags + "level queue not empty" — for a broadcast that is enqueued
    //  before the host has looked at the counters: comm.hip bvhgpu_bcast_known)
    t->bstat.reserve(64);
    if (flatten_after && n >= 1)
        flatten_tree<T>(t, a.ctr, reinterpret_cast<uint32_t*>(t->pin), (uint32_t)(ROOTKEY_OFF / 4), t->bstat.as<uint32_t>(), (uint32_t)CTR_FLAGS,
                        (n > (size_t)MID_MAX && !t->pend_persist) ? (uint32_t)(CTR_LEVEL0 + 2 * lvl_slot(level)) : (uint32_t)CTR_FLAGS,   // (persistent tier: its give-up flag IS the unfinished bit)
                        ctx->tune[BVHGPU_TUNE_FLATTEN_LAZY] != 0, inline_parts);
    else hipLaunchKernelGGL(k_publish_build<T>, dim3(1), dim3(256), 0, st, a, reinterpret_cast<uint32_t*>(t->pin));
    t->ctr_ready = true;
    t->pending_build = true; t->pend_level = level; t->pend_flatten = flatten_after;
}

template <typename T> void build_finalize(bvhgpu_tree* t) {
    if (!t->pending_build) return;
    t->pending_build = false;
    bvhgpu_ctx* ctx = t->ctx;
    hipStream_t st = ctx->stream;
    join_flat(t);
    BVH_HIP(hipStreamSynchronize(st));
    BVH_HIP(hipGetLastError());
    const size_t n = t->n;
    const size_t MID_MAX = n <= MID_SCENE_SPLIT ? (size_t)MidSmallScene<T>::MAXN : (size_t)MidLargeScene<T>::MAXN;
    uint32_t* pin = reinterpret_cast<uint32_t*>(t->pin);
    if (pin[CTR_FLAGS] & BUILD_FLAG_NONFINITE) {
        t->flattened = false;
        t->failed_gen = t->gen; t->failed_what = "NONFINITE";   // (batches already enqueued on this generation walked nothing: their waits say so)
        throw HipFail{hipErrorInvalidValue, "NONFINITE", __LINE__};
    }
    if (t->pend_persist && (pin[CTR_FLAGS] & BUILD_FLAG_PERSIST_GAVE_UP)) {
        // the persistent level tier left the tree unfinished: the same generation again with a launch per level (this tree stays with that)
        t->persist_broken = true;
        t->persist_retry_in = 64;
        const bool fl = t->pend_flatten;
        build_enqueue<T>(t, static_cast<const T*>(nullptr), n, fl, true);
        build_finalize<T>(t);
        t->redone_gen = t->gen;   // whoever traversed the optimistic result of this generation must do it again
        return;
    }
    int level = t->pend_persist ? MAXLV - 3 : t->pend_level;
    if (n > (size_t)MID_MAX && !t->pend_persist && pin[CTR_LEVEL0 + 2 * lvl_slot(level)] != 0) {
        // slow path: the level queue is not empty yet (unbalanced tree) — one more level per host round trip.  The counters
        // were zeroed behind the readback: put them back first.
        const BuildGrid<T> g(t, MID_MAX);
        const BuildArgs<T> a = make_args<T>(t, nullptr);
        BVH_HIP(hipMemcpyAsync(a.ctr, t->pin, ROOTKEY_OFF, hipMemcpyHostToDevice, st));
        t->ctr_ready = false;
        uint32_t mid2_done = pin[CTR_MID2], small_done = pin[CTR_SMALL];
        while (true) {
            if (level + 1 >= MAXLV - 2)  // recycle the slot the next level will append to
                BVH_HIP(hipMemsetAsync(a.ctr + CTR_LEVEL0 + 2 * lvl_slot(level + 1), 0, 8, st));
            run_level<T>(t, a, g, level);
            level++;
            BVH_HIP(hipMemcpyAsync(t->pin, a.ctr, ROOTKEY_OFF, hipMemcpyDeviceToHost, st));
            BVH_HIP(hipStreamSynchronize(st));
            if (pin[CTR_LEVEL0 + 2 * lvl_slot(level)] == 0) break;
        }
        run_lower_tiers<T>(t, a, g, mid2_done, small_done);
        hipLaunchKernelGGL(k_publish_build<T>, dim3(1), dim3(256), 0, st, a, reinterpret_cast<uint32_t*>(t->pin));
        t->ctr_ready = true;
        if (t->pend_flatten) {   // the optimistic one ran over an unfinished tree (and may have left stray LDS slot entries)
            BVH_HIP(hipMemsetAsync(t->wslot_node.p, 0xFF, WIDE_SLOTS * 4, st));
            BVH_HIP(hipMemsetAsync(t->slot_entry.p, 0xFF, TopCfg<T>::SLOTS * 4, st));
            flatten_tree<T>(t);
        }
        BVH_HIP(hipStreamSynchronize(st));
        BVH_HIP(hipGetLastError());
        t->redone_gen = t->gen;   // whoever traversed the optimistic result of this generation must do it again
    }
    // diagnostic: number of level-synchronous passes that had work
    int used = level;
    while (used > 0 && used - 1 < MAXLV - 2 && pin[CTR_LEVEL0 + 2 * (used - 1)] == 0) used--;
    t->levels = used;
    if (!t->pend_persist) { t->hint_levels = used > 0 ? used : 0; t->hint_n = n; }   // the next rebuild's optimistic schedule
    t->exact_only = (pin[CTR_FLAGS] & BUILD_FLAG_EMPTY_SPLIT) != 0;
    t->built = true;
}

template <typename T> void build_tree(bvhgpu_tree* t, const T* aabbs_dev, size_t n, bool flatten_after) {
    build_enqueue<T>(t, aabbs_dev, n, flatten_after, false);
    build_finalize<T>(t);
}

#ifdef BVH_PROFILE_MID
void debug_mid_prof(unsigned long long* out, bool reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mid_prof), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mid_prof), z, sizeof z); }
}
#endif

template void build_tree<float>(bvhgpu_tree*, const float*, size_t, bool);
template void build_tree<double>(bvhgpu_tree*, const double*, size_t, bool);
template void build_enqueue<float>(bvhgpu_tree*, const float*, size_t, bool, bool);
template void build_enqueue<double>(bvhgpu_tree*, const double*, size_t, bool, bool);
template void build_finalize<float>(bvhgpu_tree*);
template void build_finalize<double>(bvhgpu_tree*);

}  // namespace bvhgpu
