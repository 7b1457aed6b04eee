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
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
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
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename U> U* as() const { return reinterpret_cast<U*>(p); }
};

}  // namespace bvhgpu

struct bvhgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int n_cu = 256;
    int tune[BVHGPU_TUNE_COUNT] = {2, 32, 1, 16384, 0, 0, 1, 0};  // bvhgpu_set_tuning defaults
    // timing
    bool timing = false;
    hipEvent_t ev[8] = {};
    unsigned ev_set = 0;  // bit0 build pair recorded, bit1 flatten, bit2 traverse
    bvhgpu_timings last = {0, 0, 0, 0};
    // scratch
    bvhgpu::DevBuf upload;    // staging for host→device inputs (aabbs / rays)
    bvhgpu::DevBuf counters;  // small device counters
    void* pinned = nullptr;   // 4 KiB pinned host page for tiny D2H reads
};

struct bvhgpu_tree {
    bvhgpu_ctx* ctx = nullptr;
    int dtype = BVHGPU_F32;
    size_t n = 0;         // shapes
    size_t n_nodes = 0;   // 2n-1
    size_t n_flat = 0;    // 3n-2 (1 if n==1)
    size_t n_trav = 0;    // 2n-2 (1 if n==1)
    bool built = false;     // has nodes / shape_node (false for imported scenes)
    bool flattened = false; // has trav (+ flat if built)
    bool unfolded = false;  // trav mirrors an uploaded FlatBvh 1:1 (nav and leaf entries kept apart)
    bool ctr_ready = false; // build counters / root keys were reset by the previous build
    int levels = 0;
    // persistent device arrays
    bvhgpu::DevBuf aabbs;       // n * 6 T
    bvhgpu::DevBuf nodes;       // n_nodes * Node
    bvhgpu::DevBuf node_start;  // n_nodes * u32   (leaves before node = first sorted position)
    bvhgpu::DevBuf node_count;  // n_nodes * u32   (shapes under node)
    bvhgpu::DevBuf shape_node;  // n * u32
    bvhgpu::DevBuf flat;        // n_flat * Flat     (reference layout, for export/parity)
    bvhgpu::DevBuf trav;        // n_trav * TravNode (engine layout, what traversal reads)
    bvhgpu::DevBuf tris;        // n * 9 T triangle vertices (optional: triangle stage)
    bool has_tris = false;
    bvhgpu::DevBuf slot_entry;  // TopCfg::SLOTS * u32: traversal entry held in LDS slot s (NONE = unused slot)
    bvhgpu::DevBuf node_slot;   // n_nodes * u16: LDS slot of the node's traversal entry (SLOT_NONE = not resident)
    // build scratch (kept for rebuild)
    bvhgpu::DevBuf idx[2];      // n * u32 ping-pong permutation
    bvhgpu::DevBuf bk;          // n * u8 bucket per position
    bvhgpu::DevBuf big[2];      // Item queues of the level-synchronous tier
    bvhgpu::DevBuf mid2;        // Item queue of the workgroup tier (65..1024 shapes)
    bvhgpu::DevBuf small;       // Item queue of the wave-subtree tier
    bvhgpu::DevBuf stats[2];    // per big item: 6 x (12 keys) + 6 counts
    bvhgpu::DevBuf tile_item[2];
    bvhgpu::DevBuf tile_cnt;    // per tile 6 x u32 (counts, then exclusive offsets)
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
    bvhgpu::DevBuf blocksums;
    bvhgpu::DevBuf ctr;      // [0] pool count (u64) [1] visited [2] leaf_visits [3] device_steps [4] ray ticket
    bvhgpu::DevBuf heap_dist, heap_node;  // best-first traversal: the part of the lanes' heaps that does not fit in LDS
    uint32_t heap_cap = 48;  // ... entries per lane (doubles when a batch overflows it)
    size_t pool_cap = 0;
    bool ctr_clean = false;  // the counters were zeroed behind the previous call's readback
};

namespace bvhgpu {

// build.hip
template <typename T> void build_tree(bvhgpu_tree* t, const T* aabbs_dev, size_t n, bool flatten_after);
// flatten.hip
template <typename T> void flatten_tree(bvhgpu_tree* t);
// refit.hip
template <typename T> void refit_tree(bvhgpu_tree* t, const T* aabbs_dev);
// traverse.hip
template <typename T>
void traverse_batch(bvhgpu_tree* t, const typename Traits<T>::Ray* rays_dev, size_t n_rays, unsigned flags,
                    bvhgpu_hits* h);
template <typename T>
void nearest_batch(bvhgpu_tree* t, const T* points_dev, size_t n, int kind, uint32_t* out_shape_dev, T* out_dist_dev);
template <typename T>
void ray_triangle_pairs(bvhgpu_ctx* ctx, const typename Traits<T>::Ray* rays_dev, const T* tris_dev, size_t n, T* out_dev);
template <typename T>
void rays_new(bvhgpu_ctx* ctx, const T* origins_dev, const T* dirs_dev, size_t n, typename Traits<T>::Ray* out_dev);
template <typename T>
void gen_primary(bvhgpu_ctx* ctx, const float cam[14], uint32_t width, uint32_t height, uint64_t first, size_t n,
                 typename Traits<T>::Ray* out_dev);
void gen_rays_f32(bvhgpu_ctx* ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f32* out_dev);
void gen_rays_f64(bvhgpu_ctx* ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f64* out_dev);

}  // namespace bvhgpu
