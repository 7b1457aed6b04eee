// engine.hpp — host-side objects behind the opaque C handles, and the per-phase entry points
// implemented in build.hip / flatten.hip / traverse.hip.
#pragma once

#include <string>
#include <vector>

#include "common.hpp"

namespace bvhgpu {

// RAII-less device buffer that only ever grows (no allocation on the steady-state hot loop)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool reserve(size_t bytes) {   // true: a new (uninitialised) allocation was made
        if (bytes <= cap) return false;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            throw HipFail{e, "hipMalloc", __LINE__};
        }
        cap = want;
        return true;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename U> U* as() const { return reinterpret_cast<U*>(p); }
};

}  // namespace bvhgpu

struct bvhgpu_comm;
struct bvhgpu_hits;
namespace bvhgpu {
// bvhgpu_traverse_host_*: a batch whose rays start in host memory and whose CSR ends there, walked as `chunks` ordinary asynchronous
// batches so that the upload of one chunk (side stream), the walk of the previous one (main stream) and the download of the offsets of
// the one before (down stream) overlap
struct HostBatch {
    static constexpr int MAX_CHUNKS = 16;
    bvhgpu_hits* hits[MAX_CHUNKS] = {};
    hipEvent_t ev_up[MAX_CHUNKS] = {}, ev_done[MAX_CHUNKS] = {};
    hipStream_t up = nullptr, up2 = nullptr, down = nullptr;
    hipEvent_t ev_main = nullptr, ev_aabbs = nullptr, ev_up2[MAX_CHUNKS] = {};
    DevBuf indices;   // the batch's index lists in one piece, as far as the caller's buffer reaches (at most IDX_STAGE_MAX entries)
    static constexpr size_t IDX_STAGE_MAX = (size_t)1 << 26;
    size_t idx_stage = 0;      // entries of `indices` this batch may fill
    uint64_t guess = 0;        // entries of `indices` whose download was enqueued before the batch's total was known (the previous batch's total)
    DevBuf od;        // origins + directions of the batch as uploaded (2 x n_rays x 3 T), or nothing when the caller hands over Ray structs
    DevBuf rays;      // n_rays x Ray
    DevBuf offsets;   // n_rays + 1 u32: the whole batch's CSR offsets (chunks rebased)
    size_t n_rays = 0, r0[MAX_CHUNKS + 1] = {};
    int chunks = 0;
    uint64_t total = 0;
    bool fetched = false;
    bool zero_copy_in = false;   // the device reads the caller's (pinned) ray arrays itself: Ray::new straight out of host memory, no staging copy
    uint32_t* offsets_host = nullptr;   // device-visible address of the caller's (pinned) offsets / indices arrays: written by the device, no download
    uint32_t* indices_host = nullptr;
    bool od6 = false;       // ... side by side in ONE array of n_rays x 6 (BVHGPU_TRAVERSE_RAYS_OD6)
    bool with_od = false;   // the batch came as origins + directions (Ray::new on the device) rather than as Ray structs
};
}

#ifndef BVH_FLATTEN_INLINE_DEFAULT
#define BVH_FLATTEN_INLINE_DEFAULT 1
#endif
struct bvhgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int n_cu = 256;
    int tune[BVHGPU_TUNE_COUNT] = {3, -1, -1, 16384, 0, 0, 1, 0, 0, 0, 0, 0, -1, 1, 1, 1, 0, 0, 256, 2, 0, BVH_FLATTEN_INLINE_DEFAULT};  // bvhgpu_set_tuning defaults
    // timing
    bool timing = false;
    hipEvent_t ev[8] = {};
    unsigned ev_set = 0;  // bit0 build pair recorded, bit1 flatten, bit2 traverse
    bvhgpu_timings last = {0, 0, 0, 0};
    // scratch
    bvhgpu::DevBuf upload;    // staging for host→device inputs (aabbs / rays)
    bvhgpu::DevBuf counters;  // small device counters
    void* pinned = nullptr;   // 4 KiB pinned host page for tiny D2H reads
    bvhgpu::HostBatch* host = nullptr;   // state of bvhgpu_traverse_host_* (capi.hip), made on first use
    hipStream_t side = nullptr;   // second stream of the ctx (created on first use): work that may run BESIDE the main chain — the
                                  // item filter of a batch whose tree is still building (traverse.hip k_wide_items)
};

struct bvhgpu_tree {
    bvhgpu_ctx* ctx = nullptr;
    int dtype = BVHGPU_F32;
    size_t n = 0;         // shapes
    size_t n_nodes = 0;   // 2n-1
    size_t n_flat = 0;    // 3n-2 (1 if n==1)
    size_t n_trav = 0;    // 2n-2 (1 if n==1)
    bool built = false;     // has nodes / shape_node (false for imported scenes)
    bool flattened = false; // has trav (+ flat if built) — or owes them: see lazy_flat
    bool lazy_flat = false; // the flatten of this generation wrote the wide walk's arrays only (BVHGPU_TUNE_FLATTEN_LAZY): flat / trav /
                            // slot_entry are written by ensure_flat_arrays() the first time something reads them
    int lazy_parts = 0;     // ... which of them (flatten.hip FLATTEN_FLAT | FLATTEN_TRAV; BVHGPU_TUNE_FLATTEN_LAZY = 3 owes the binary array only)
    bool flat_beside = false;   // part 1 of this generation's flatten (flat / trav / slot_entry) is in flight on the ctx's SIDE stream
    hipEvent_t ev_flat0 = nullptr, ev_flat = nullptr;   // (BVHGPU_TUNE_FLATTEN_LAZY = 2): main → side, side → main (join_flat)
    bool unfolded = false;  // trav mirrors an uploaded FlatBvh 1:1 (nav and leaf entries kept apart)
    bool ctr_ready = false; // build counters / root keys were reset by the previous build
    bool pending_build = false;  // build_enqueue ran, build_finalize has not (asynchronous entry points)
    bool pend_flatten = false;   // ... and the flatten was enqueued behind it
    uint64_t gen = 0;            // generation of the tree's contents: every build_enqueue / scene import / broadcast receive starts a new one
    uint64_t redone_gen = 0;     // the generation whose build_finalize had to finish the tree on the slow path (0: none): batches that
                                 // were enqueued on that generation before the finalize walked an unfinished tree and are replayed by
                                 // bvhgpu_hits_wait — by EVERY result object that recorded it, whoever finalized the build first
    uint64_t bcast_gen = 0;      // the generation that bvhgpu_bcast_known sent before its build was finalized (comm.hip)
    uint64_t failed_gen = 0;     // the generation whose build_finalize / recv_finalize found nothing usable (NaN / inf input, a root without a
    const char* failed_what = nullptr;   // valid tree): WHOEVER consumed that error first (bvhgpu_tree_wait, a rebuild, another result object's
                                 // wait), every asynchronous batch that was enqueued on that generation returns it from its own wait
    bool pending_recv = false;   // a broadcast was received on the stream; its status header (t->pin_recv) has not been looked at yet
    void* pin_recv = nullptr;    // 64 B of pinned host memory: the received broadcast header
    bvhgpu_comm* recv_comm = nullptr;
    bvhgpu::DevBuf bstat;        // 64 B: build status word for the device-side broadcast header (written by k_flatten's publishing block)
    std::vector<bvhgpu_hits*> waiters;   // asynchronous batches enqueued on this tree that bvhgpu_hits_wait has not completed yet
    hipEvent_t ev_top = nullptr;         // recorded by build_enqueue behind the level pass that splits tree level 3: from then on the
    uint64_t ev_top_gen = 0;             // BvhNode records of tree levels 0..3 of generation ev_top_gen are final (if the level tier wrote
                                         // them: counter CTR_TOPMASK) — what the walk's item filter needs, 100+ µs before the tree is complete
    bool exact_only = false;     // some split had no SAH winner (empty child bounds): a child box is not the join of its
                                 // grandchildren, so traversal must test every ancestor (binary walk only)
    int pend_level = 0;
    bool pend_persist = false;   // the build in flight ran the level tier's lower passes as one persistent launch (build.hip k_level_xcd)
    bool persist_broken = false; // ... which gave up on this tree (its workgroups were not resident together): a launch per level for the next
    int persist_retry_in = 0;    //     persist_retry_in builds, then the persistent tier is tried again
    int levels = 0;
    int hint_levels = 0;         // level-synchronous passes the previous build of hint_n shapes needed
    size_t hint_n = 0;
    void* pin = nullptr;         // pinned host copy of the build counters (1 KiB), one per tree so that builds can be in flight
    // persistent device arrays
    bvhgpu::DevBuf aabbs;       // n * 6 T
    bvhgpu::DevBuf nodes;       // n_nodes * Node
    bvhgpu::DevBuf node_start;  // n_nodes * u32   (leaves before node = first sorted position)
    bvhgpu::DevBuf node_count;  // n_nodes * u32   (shapes under node)
    bvhgpu::DevBuf shape_node;  // n * u32
    bvhgpu::DevBuf flat;        // n_flat * Flat     (reference layout, for export/parity)
    bvhgpu::DevBuf trav;        // n_trav * TravNode (engine layout, what traversal reads)
    bvhgpu::DevBuf wide;        // n_nodes * WideNode: the four grandchildren of every inner node (wide walk, traverse.hip)
    bvhgpu::DevBuf wslot_node;  // WideCfg::SLOTS * u32: tree node held in 4-ary heap slot s of the LDS-resident top (NONE = none)
    bool has_wide = false;
    // f64 trees: the same wide nodes as f32 boxes that CONTAIN the f64 ones (rounded outward and grown by GUIDE_GROW x the scene's largest
    // |coordinate|): index batches are walked over these with f32 rays and only leaf candidates are tested in f64 (traverse.hip "guide walk")
    bvhgpu::DevBuf wide_guide;  // n_nodes * WideNode<float>
    bvhgpu::DevBuf guide_info;  // float[4]: [0] = S, the largest |coordinate| of the root's child boxes
    bool has_guide = false;
    bvhgpu::DevBuf tris;        // n * 9 T triangle vertices (optional: triangle stage)
    bool has_tris = false;
    bvhgpu::DevBuf slot_entry;  // TopCfg::SLOTS * u32: traversal entry held in LDS slot s (NONE = unused slot)
    bvhgpu::DevBuf node_slot;   // n_nodes * u16: LDS slot of the node's traversal entry (SLOT_NONE = not resident)
    // build scratch (kept for rebuild)
    bvhgpu::DevBuf idx[2];      // n * u32 ping-pong permutation
    bvhgpu::DevBuf bk;          // 2 n * u8 bucket per position (two consecutive levels)
    bvhgpu::DevBuf lvbuf;       // level tier, one launch per level: rotating tile maps / tile counts / statistics (build.hip LevelLayout)
    bvhgpu::DevBuf xbar;        // persistent level tier: barrier words of its eight workgroup groups (2 KB)
    bvhgpu::DevBuf big[2];      // Item queues of the level-synchronous tier
    bvhgpu::DevBuf mid2;        // Item queue of the workgroup tier (65..1024 shapes)
    bvhgpu::DevBuf small;       // Item queue of the wave-subtree tier
    bvhgpu::DevBuf stats[2];    // per big item: 6 x (12 keys) + 6 counts
    bvhgpu::DevBuf tile_item[2];
    bvhgpu::DevBuf tile_cnt;    // per tile 6 x u32 (counts, then exclusive offsets)
    bvhgpu::DevBuf chunk_cnt;   // two-launch level schedule: the same counts per block of 256 tile ids (build.hip BuildArgs::chunk_cnt)
    bvhgpu::DevBuf ctr;         // counters
    bvhgpu::DevBuf refit_seg;   // refit: complete binary tree of joins over the sorted positions (2 * n_pad boxes)
};

struct bvhgpu_hits {
    bvhgpu_ctx* ctx = nullptr;
    int dtype = BVHGPU_F32;
    size_t n_rays = 0;
    uint64_t total = 0;
    unsigned flags = 0;
    bvhgpu_traverse_stats stats = {0, 0, 0, 0, 0};
    bvhgpu::DevBuf counts;   // n_rays+1 u32 (counts, scanned in place into offsets)
    bvhgpu::DevBuf offsets;  // n_rays+1 u32
    bvhgpu::DevBuf pool;     // hit records (ray, k, shape)
    bvhgpu::DevBuf pool_t;   // 2 T per record
    bvhgpu::DevBuf indices;  // total u32
    bvhgpu::DevBuf tslice;   // total * 2 T
    bvhgpu::DevBuf isect;    // total * 3 T: Intersection{distance,u,v} per candidate (TRIANGLES)
    bvhgpu::DevBuf closest;  // n_rays * 3 T (CLOSEST)
    bvhgpu::DevBuf closest_prim;  // n_rays u32
    bvhgpu::DevBuf closest_key;   // n_rays u64: CLOSEST batches walked as items (traverse.hip WalkOut::closest_key), all-ones between batches
    bool ckey_clean = false;
    bvhgpu::DevBuf blocksums;
    bvhgpu::DevBuf scan_sums;    // wide walk: two sets of hits per scan block (k_scan_final's input instead of a reduce pass), kept zero
    bvhgpu::DevBuf ctr;      // [0] pool count (u64) [1] visited [2] leaf_visits [3] device_steps [4] ray ticket
    bvhgpu::DevBuf heap_dist, heap_node;  // best-first traversal: the part of the lanes' heaps that does not fit in LDS
    uint32_t heap_cap = 48;  // ... entries per lane (doubles when a batch overflows it)
    size_t pool_cap = 0;
    size_t idx_cap = 0;      // capacity of indices[] in entries (>= pool_cap; staged output sizes it by the hit total)
    bvhgpu::DevBuf raybuf;   // staged output of the wide walk: 2^shift shape indices per ray (traverse.hip WalkOut::raybuf)
    bool pend_staged = false;
    std::string walk_kernel;  // the kernel the last batch was handed to, as rocprofv3 spells it (bvhgpu_hits_walk_kernel)
    bool pend_rec8 = false;   // the batch in flight writes pair records (8 bytes per hit: traverse.hip report_pair)
    bool pend_guide = false, no_guide = false; // guide walk in the batch in flight / the batch is being replayed in f64 (a ray was out of the guide's range)
    uint32_t guide_backoff = 0, guide_skip = 0; // f64 index batches that skip the guide after such a replay: 1, 2, 4 … 64 on consecutive failures / still to skip
    bool ctr_clean = false;  // the counters were zeroed behind the previous call's readback
    int ctr_set = 0;         // which of the two counter sets the next batch uses
    int bsum_set = 0;        // likewise for the wide walk's scan-block sums
    // wide walk
    bvhgpu::DevBuf wcounts;   // n_rays+1 u32: hit count | item mask << 28, all-zero between batches
    bvhgpu::DevBuf ray_mask;  // n_rays u16: items of the ray that reported hits (valid for rays with hits)
    bvhgpu::DevBuf ray_items; // n_rays u32: the same set while the walk collects it (atomicOr), all-zero between batches
    bvhgpu::DevBuf witems;    // live items (ray << 5 | j) of the batch, from k_wide_items
    bvhgpu::DevBuf item_cnt;  // per item: hits (valid where the ray's mask has the item's bit)
    bvhgpu::DevBuf wstack;    // the part of the lanes' stacks that does not fit in LDS
    bool wcounts_clean = false;
    bool bs_clean = false;       // blocksums are all-zero (the wide walk adds into them)
    bool force_binary = false;   // the wide walk overflowed a lane's stack on this batch: replay with the binary walk
    // the batch in flight (traverse_enqueue → traverse_check): what a replay needs
    void* pin = nullptr;         // 64 B of pinned host memory: the 8 walk / scan counters of the last batch
    bvhgpu_tree* pend_tree = nullptr;
    const void* pend_rays = nullptr;
    bool pend_wide = false, pend_unfolded = false, pend_async = false;
    int pend_attempts = 0;
    // asynchronous batches: what bvhgpu_hits_wait needs to decide whether the optimistic walk has to be replayed
    bvhgpu_tree* wait_tree = nullptr;   // the tree whose `waiters` list holds this object (NULL: none)
    uint64_t pend_gen = 0;              // generation of the tree the batch was enqueued on
    bool pend_on_pending = false;       // ... and that generation was not finalized yet at that moment
    hipEvent_t ev_items = nullptr;      // the early item filter of this batch has finished (side stream → main stream)
    bvhgpu::DevBuf wg_items;            // per workgroup of the wide walk: {items at the front, items at the back} of its list region, written by
                                        // the early filter (front == NONE: the workgroup filters its rays itself)
    uint32_t replays = 0;               // times bvhgpu_hits_wait had to enqueue the asynchronous batch again
    int deferred_rc = 0;                // status of a completion that ran on behalf of another call (rebuild / destroy of the tree)
    std::string deferred_err;
};

namespace bvhgpu {

// build.hip
template <typename T> void build_tree(bvhgpu_tree* t, const T* aabbs_dev, size_t n, bool flatten_after);
template <typename T> void build_enqueue(bvhgpu_tree* t, const T* aabbs_dev, size_t n, bool flatten_after, bool redo = false);
template <typename T> void build_finalize(bvhgpu_tree* t);
// flatten.hip
// pub_*: also publish + reset the builder's counters (build_enqueue's last launch); see k_flatten
// bstat (with pub_*): device-side status word = pub_ctr[flags_idx] | (pub_ctr[level_idx] != 0 ? BSTAT_UNFINISHED : 0), level_idx == flags_idx: no level tier
// wide_only: write the wide walk's arrays only and leave flat / trav / slot_entry to ensure_flat_arrays (ignored where the tree has no wide nodes)
// inline_parts: the parts the builder's wave tier has already written for every node of at most SMALL_MAX shapes (build.hip k_small, flatten_node.hpp)
template <typename T> void flatten_tree(bvhgpu_tree* t, uint32_t* pub_ctr = nullptr, uint32_t* pub_host = nullptr, uint32_t pub_words = 0,
                                        uint32_t* bstat = nullptr, uint32_t flags_idx = 0, uint32_t level_idx = 0, bool wide_only = false,
                                        uint32_t inline_parts = 0);
struct FlattenPlan { int parts = 0; bool with_wide = false, with_guide = false; };
template <typename T> FlattenPlan flatten_plan(bvhgpu_tree* t, bool wide_only);
// the FlatNode array, the folded binary array and the binary walk's LDS slot table of a tree whose flatten was lazy: enqueued on the
// tree's stream now (a no-op for every other tree).  Everything that reads t->flat / t->trav / t->slot_entry calls this first.
void ensure_flat_arrays(bvhgpu_tree* t);
// BVHGPU_TUNE_FLATTEN_LAZY = 2: the main stream waits for the part of the flatten that runs on the side stream (a no-op otherwise).  Called
// behind the walk a batch enqueues (so that the batch's wait covers it), and before anything reads those arrays or overwrites what they are made from.
void join_flat(bvhgpu_tree* t);
constexpr uint32_t BSTAT_NONFINITE = 1u, BSTAT_EMPTY_SPLIT = 2u;   // = build.hip BUILD_FLAG_*
constexpr uint32_t BSTAT_UNFINISHED = 0x100u;
constexpr int BUILD_CTR_TOPMASK = 5;   // u32 slot of the build counters: bit h = the level tier wrote the BvhNode of heap number h (h < 16)                      // the optimistic schedule left nodes in the level queue
template <typename T> void wide_from_trav(bvhgpu_tree* t);   // wide nodes + their LDS slot table from trav + slot_entry
// comm.hip: completes a broadcast that was received on the stream (reads the status header; throws RECV_* on a bad one)
void recv_finalize(bvhgpu_tree* t);
// capi.hip: completes the asynchronous batches still in flight on a tree whose arrays are about to be overwritten or freed
void settle_waiters(bvhgpu_tree* t);
// refit.hip
template <typename T> void refit_tree(bvhgpu_tree* t, const T* aabbs_dev);
// traverse.hip
template <typename T>
void traverse_batch(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, unsigned flags,
                    bvhgpu_hits* h);
template <typename T>
void traverse_enqueue(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, unsigned flags,
                      bvhgpu_hits* h);
bool traverse_check(bvhgpu_hits* h);   // after a stream synchronise: false = enqueue again
template <typename T>
void nearest_batch(bvhgpu_tree* t, const T* points_dev, size_t n, int kind, uint32_t* out_shape_dev, T* out_dist_dev);
template <typename T>
void ray_triangle_pairs(bvhgpu_ctx* ctx, const typename Traits<T>::Ray* rays_dev, const T* tris_dev, size_t n, T* out_dev);
template <typename T>
void rays_new(bvhgpu_ctx* ctx, const T* origins_dev, const T* dirs_dev, size_t n, typename Traits<T>::Ray* out_dev, hipStream_t st = nullptr,
              unsigned max_blocks = 0, unsigned stride = 3);
// out[i] = out[0] + offs[i], i = 1..n_rays; idx_all_dev != NULL: the chunk's index list copied to idx_all_dev[out[0] ..) as far as idx_cap entries reach
// out_host (device-visible pinned host memory, or NULL): the same values stored there as well; idx_all (device or such host memory)
void offsets_rebase(hipStream_t st, const uint32_t* offs_dev, size_t n_rays, uint32_t* out_dev, uint32_t* out_host = nullptr,
                    const uint32_t* idx_dev = nullptr, uint32_t* idx_all = nullptr, size_t idx_cap = 0, bool first = false);
void copy16(hipStream_t st, const void* src, void* dst, size_t bytes);   // a copy KERNEL (src / dst may be device-visible host memory)
template <typename T>
void gen_primary(bvhgpu_ctx* ctx, const float cam[14], uint32_t width, uint32_t height, uint64_t first, size_t n,
                 typename Traits<T>::Ray* out_dev);
void gen_rays_f32(bvhgpu_ctx* ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f32* out_dev);
void gen_rays_f64(bvhgpu_ctx* ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f64* out_dev);

}  // namespace bvhgpu
