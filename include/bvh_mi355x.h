/*
 * bvh_mi355x.h — C ABI of libbvh_mi355x.so: the MI355X (gfx950 / CDNA4) engine for the hot path
 * of the Rust crate `bvh` 0.12.0 (svenstaro/bvh): SAH build → flatten → batched ray traversal.
 *
 * The reference has no FFI of its own (pure Rust); its extension points for this path are the
 * `BoundingHierarchy` trait (src/bounding_hierarchy.rs:89-336), `Bvh::flatten_custom`
 * (src/flat_bvh.rs:240-251) and `build_with_executor` (src/bvh/bvh_impl.rs:53-56).  Each entry
 * point below names the reference interface it replaces (paths relative to the crate root).  The
 * Rust-side binding a maintainer would add (a `GpuBvh` type that implements `BoundingHierarchy`
 * over these symbols) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; every function returns a bvhgpu_status (0 = OK) and never throws or
 *    aborts across the boundary; bvhgpu_last_error(ctx) gives the text of the last failure.
 *  - "mem" arguments say where a caller buffer lives: BVHGPU_HOST (pageable or pinned host
 *    memory) or BVHGPU_DEVICE (HBM of the ctx's GPU, e.g. a torch tensor's data_ptr()).
 *  - all work is enqueued on the ctx's HIP stream; host-visible results are complete when the
 *    call returns (the call synchronises the stream), device-side results are ordered on the stream.
 *  - a ctx is not internally locked: one ctx per host thread.  Trees are immutable after
 *    build/flatten and may be traversed from several ctxs of the same device concurrently.
 *  - Semantics are bit-exact with the reference for indices/topology: NUM_BUCKETS
 *    = 6 (src/bvh/bucket.rs:5), pre-order node placement (src/bvh/bvh_node.rs:138-142), stable
 *    bucket-major index rewrite (:250-272), strict-< first-wins SAH argmin (:239), surface_area =
 *    2*dot(size,size) (src/aabb/aabb_impl.rs:551-554), NaN-in-slab = miss
 *    (src/ray/intersect_default.rs:22-28).
 *  - Input contract of the builders: the reference panics on a NaN (or infinite) centroid — `to_usize().unwrap()`,
 *    src/bvh/bvh_node.rs:214-217.  bvhgpu_build_* / rebuild_* detect NaN / ±inf in the shape AABBs (and finite boxes whose
 *    centroid extent overflows) on the device and return BVHGPU_INVALID_ARG without building anything; a single shape
 *    is never bucketed and is accepted as in the reference.
 */
#ifndef BVH_MI355X_H
#define BVH_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BVHGPU_ABI_VERSION 7 /* 7: bvhgpu_host_alloc/free/register/unregister, bvhgpu_traverse_host_*, bvhgpu_build_traverse_host_*, bvhgpu_traverse_host_indices, BVHGPU_TUNE_COUNT 20 (slots 17 = HOST_CHUNKS, 18 = WIDE_MIN_RAYS_PER_WG, 19 = HOST_ZERO_COPY).  6: BVHGPU_TUNE_COUNT 17 (slots 15 = FLATTEN_LAZY, 16 = BUILD_LEVEL_PERSIST), bvhgpu_hits_walk_kernel.  5: bvhgpu_rccl_info.  4: BVHGPU_TUNE_COUNT 15 (slot 14 = WIDE_F64_GUIDE), bvhgpu_hits_walk_info.  3: BVHGPU_REBROADCAST, broadcast status header, scene blob BVH6 (exact_only), BVHGPU_TUNE_COUNT 14 (slots 11 = WIDE_EARLY_ITEMS, 12 = WIDE_STAGE_SHIFT, 13 = WIDE_REC8), BVHGPU_TRAVERSE_RAYS_READY, bvhgpu_device_alloc/free/copy */
#define BVHGPU_NONE 0xFFFFFFFFu /* u32::MAX marker (flat_bvh.rs:51-53, :124, :137) */

typedef enum {
    BVHGPU_OK = 0,
    BVHGPU_INVALID_ARG = 1,
    BVHGPU_HIP_ERROR = 2,
    BVHGPU_OOM = 3,
    BVHGPU_OVERFLOW = 4,   /* > (2^32-2)/3 shapes (flat_bvh.rs:136 would truncate silently) or > 2^32-1 hits */
    BVHGPU_NO_DEVICE = 5,
    BVHGPU_DTYPE_MISMATCH = 6,
    BVHGPU_NOT_FLATTENED = 7,
    BVHGPU_RCCL_ERROR = 8, /* an RCCL call of the multi-GPU broadcast failed (bvhgpu_last_error has ncclGetErrorString), or librccl
                              could not be loaded (it is dlopen'ed on first use: single-GPU consumers do not need it) */
    BVHGPU_REBROADCAST = 9 /* returned by a wait (bvhgpu_tree_wait / bvhgpu_hits_wait / any call that inspects the tree) on EVERY rank
                              when bvhgpu_bcast_known sent a tree whose asynchronous build then turned out to need the slow path
                              (unbalanced tree on a first build): the root's tree and results are complete, the peers received
                              nothing usable — every rank calls bvhgpu_bcast_known again */
} bvhgpu_status;

typedef enum { BVHGPU_F32 = 0, BVHGPU_F64 = 1 } bvhgpu_dtype;
typedef enum { BVHGPU_HOST = 0, BVHGPU_DEVICE = 1 } bvhgpu_mem;

/* traversal flags */
#define BVHGPU_TRAVERSE_T_SLICE 1u /* also return (tmin,tmax) per hit: Ray::intersection_slice_for_aabb (ray_impl.rs:118-145) */
#define BVHGPU_TRAVERSE_STATS 2u   /* also count reference-equivalent loop iterations (flat_bvh.rs:408).  Exact for every tree whose
                                      leaves' navigator boxes equal the shapes' AABBs — all but trees with a split that had no SAH
                                      winner (empty child bounds, bvh_node.rs:225-230): there the leaf-entry visits behind an empty
                                      navigator box are not counted (the hit lists are still the reference's) */
#define BVHGPU_TRAVERSE_TRIANGLES 4u /* also run Ray::intersects_triangle (ray_impl.rs:154-213) on every returned shape, as the
                                        reference's harness does after traverse (testbase.rs:826-836): Intersection{distance,u,v} per hit */
#define BVHGPU_TRAVERSE_CLOSEST 8u   /* triangle stage fused into the walk, no CSR: per ray the candidate with the smallest
                                        Intersection.distance (first one on ties) and its shape index */

#define BVHGPU_TRAVERSE_NEAREST_FIRST 32u  /* per-ray order of Bvh::nearest_child_traverse_iterator (bvh_impl.rs:184-190,
                                              child_distance_traverse.rs) instead of flat-array order */
#define BVHGPU_TRAVERSE_FARTHEST_FIRST 64u /* ... of Bvh::farthest_child_traverse_iterator (bvh_impl.rs:206-212) */
#define BVHGPU_TRAVERSE_BEST_FIRST 128u     /* with NEAREST_FIRST / FARTHEST_FIRST: the per-ray order of Bvh::nearest_traverse_iterator /
                                              farthest_traverse_iterator (bvh_impl.rs:145-176) = DistanceTraverseIterator
                                              (distance_traverse.rs:40-158), a best-first walk driven by a BinaryHeap, instead of
                                              the child-ordered depth-first iterator */
#define BVHGPU_TRAVERSE_RAYS_READY 256u /* hint for bvhgpu_traverse_async_*: the rays were complete before the tree's asynchronous rebuild was
                                          enqueued (resident ray buffers of a frame loop).  The engine may then read them while the build
                                          is still running — the wide walk's per-ray item filter runs beside the build on a second
                                          stream instead of in front of the walk.  Results never depend on it */
#define BVHGPU_TRAVERSE_RAYS_OD6 512u /* bvhgpu_traverse_host_* / bvhgpu_build_traverse_host_* only: `origins` points to n_rays x 6 T — origin xyz, direction xyz
                                        per ray, the arguments of Ray::new side by side — and `directions` is ignored.  One transfer per chunk
                                        instead of two: the copy engines idle 10 - 20 µs between two transfers */
#define BVHGPU_TRAVERSE_COHERENT 16u /* hint: neighbouring rays are similar (primary rays).  Large whole-ray batches then hand their hits over
                                        through per-ray slots instead of pool records (BVHGPU_TUNE_WIDE_STAGE_SHIFT); results never depend on it */

/* ---- POD layouts (little-endian, natural alignment, no packing pragmas) ---- */

/* Aabb<T,3> (aabb_impl.rs:10-16) is passed as 6 scalars: min x,y,z, max x,y,z. */

/* enum BvhNode<T,3> (bvh_node.rs:21-47) as a POD.
 * Node: shape == BVHGPU_NONE; l,r = child_l_index, child_r_index; l_min..r_max = child AABBs.
 * Leaf: shape = shape_index; l = r = BVHGPU_NONE; AABB fields are zero. */
typedef struct { float l_min[3], l_max[3], r_min[3], r_max[3]; uint32_t parent, l, r, shape; } bvhgpu_node_f32;  /*  64 B */
typedef struct { double l_min[3], l_max[3], r_min[3], r_max[3]; uint32_t parent, l, r, shape; } bvhgpu_node_f64; /* 112 B */

/* struct FlatNode<T,3> (flat_bvh.rs:17-46), same field order.  Leaf entries carry
 * Aabb::empty() = (+inf,-inf) exactly like flat_bvh.rs:133. */
typedef struct { float min[3], max[3]; uint32_t entry, exit, shape; } bvhgpu_flat_f32;                 /* 36 B */
typedef struct { double min[3], max[3]; uint32_t entry, exit, shape, _pad; } bvhgpu_flat_f64;          /* 64 B */

/* struct Ray<T,3> (ray_impl.rs:17-29), same field order: origin, direction (normalised), inv_direction. */
typedef struct { float o[3], d[3], inv[3]; } bvhgpu_ray_f32;   /* 36 B */
typedef struct { double o[3], d[3], inv[3]; } bvhgpu_ray_f64;  /* 72 B */

typedef struct bvhgpu_ctx bvhgpu_ctx;
typedef struct bvhgpu_tree bvhgpu_tree;
typedef struct bvhgpu_hits bvhgpu_hits;

typedef struct {
    uint64_t hits;        /* total shapes returned                                                  */
    uint64_t visited;     /* reference loop iterations (flat_bvh.rs:408) = slab tests; STATS flag   */
    uint64_t leaf_visits; /* of which leaf entries (second test of a shape AABB, :411-418); STATS   */
    uint64_t device_steps;/* node visits this engine actually performed (folded layout); STATS      */
    uint64_t wave_steps;  /* wavefront iterations of the walk loop: device_steps / (64 * wave_steps) = lane utilisation */
} bvhgpu_traverse_stats;

/* ---- context ---- */
int bvhgpu_abi_version(void);
int bvhgpu_device_count(int *out);
const char *bvhgpu_status_string(int status);
/* One ctx = one GPU + one stream + scratch.  stream == NULL → the ctx creates its own stream;
 * otherwise `stream` is a hipStream_t owned by the caller (e.g. torch's current stream). */
int bvhgpu_create(int device, void *stream, bvhgpu_ctx **out);
void bvhgpu_destroy(bvhgpu_ctx *ctx);
const char *bvhgpu_last_error(const bvhgpu_ctx *ctx);
int bvhgpu_synchronize(bvhgpu_ctx *ctx);
void *bvhgpu_stream(bvhgpu_ctx *ctx); /* the hipStream_t work is enqueued on */

/* ---- device memory for hosts without HIP bindings (a Rust or C caller that keeps rays / AABBs resident in HBM, the
 * BVHGPU_DEVICE form of every `mem` argument): hipMalloc / hipFree / a synchronous hipMemcpy on the ctx's device.
 * Memory from any other allocator of the same device (hipMalloc, a torch tensor) is just as good. ---- */
int bvhgpu_device_alloc(bvhgpu_ctx *ctx, size_t bytes, void **out);
int bvhgpu_device_free(bvhgpu_ctx *ctx, void *ptr);
int bvhgpu_device_copy(bvhgpu_ctx *ctx, void *dst, int dst_mem, const void *src, int src_mem, size_t bytes);

/* ---- pinned host memory (ABI 7).  Every BVHGPU_HOST argument may be ordinary (pageable) memory; memory from
 * bvhgpu_host_alloc — or a range the caller owns and has announced with bvhgpu_host_register, e.g. the allocation behind a
 * Vec that lives as long as the scene — is read and written by the DMA engines directly (≈ 55 GB/s on the PCIe 5 link instead of
 * ≈ 30 through the runtime's staging copies) and lets the asynchronous entry points return before the copy has happened.
 * hipHostMalloc / hipHostFree / hipHostRegister / hipHostUnregister on the ctx's device. ---- */
int bvhgpu_host_alloc(bvhgpu_ctx *ctx, size_t bytes, void **out);
int bvhgpu_host_free(bvhgpu_ctx *ctx, void *ptr);
int bvhgpu_host_register(bvhgpu_ctx *ctx, void *ptr, size_t bytes);
int bvhgpu_host_unregister(bvhgpu_ctx *ctx, void *ptr);

/* ---- build: replaces Bvh::build / Bvh::build_par / build_with_executor (bvh_impl.rs:40-96,
 * bounding_hierarchy.rs:158-177) once the caller has gathered shape.aabb() for every shape
 * (aabb_impl.rs:28-56) into `aabbs` = n x [min xyz, max xyz].  n == 0 → valid empty tree
 * (bvh_impl.rs:57-59).  The caller keeps `aabbs`; the tree owns its own HBM copy. ---- */
int bvhgpu_build_f32(bvhgpu_ctx *ctx, const float *aabbs, size_t n, int mem, bvhgpu_tree **out);
int bvhgpu_build_f64(bvhgpu_ctx *ctx, const double *aabbs, size_t n, int mem, bvhgpu_tree **out);
/* Rebuild in place (same dtype, n <= capacity of the first build): no allocation on the hot loop.  A different shape count
 * drops the triangle vertices of bvhgpu_tree_set_triangles (one triangle per shape); with the same count they are kept and
 * are the caller's to refresh. */
int bvhgpu_rebuild_f32(bvhgpu_tree *tree, const float *aabbs, size_t n, int mem);
int bvhgpu_rebuild_f64(bvhgpu_tree *tree, const double *aabbs, size_t n, int mem);
/* FlatBvh::build (flat_bvh.rs:328-331) = Bvh::build + Bvh::flatten in ONE call (one host round trip): the tree is
 * both built and flattened on return. */
int bvhgpu_build_flat_f32(bvhgpu_ctx *ctx, const float *aabbs, size_t n, int mem, bvhgpu_tree **out);
int bvhgpu_build_flat_f64(bvhgpu_ctx *ctx, const double *aabbs, size_t n, int mem, bvhgpu_tree **out);
int bvhgpu_rebuild_flat_f32(bvhgpu_tree *tree, const float *aabbs, size_t n, int mem);
int bvhgpu_rebuild_flat_f64(bvhgpu_tree *tree, const double *aabbs, size_t n, int mem);
/* The shapes moved but are the same shapes: keep the topology, recompute every child AABB from the new shape AABBs —
 * Bvh::fix_aabbs_ascending (optimization.rs:355-391: child boxes = children's get_node_aabb, bvh_node.rs:616-625)
 * applied to the whole tree.  n must equal the tree's shape count; the tree must have been built here.  If the tree
 * is flattened its flat / traversal arrays are regenerated.  The result is consistent and tight (the properties
 * the reference asserts after update_shapes, optimization.rs tests); refit with the AABBs of the build reproduces
 * the built tree bit for bit.  Bvh::update_shapes itself (optimization.rs:337-352: remove + re-insert, one shape at a
 * time, topology changes) has no device counterpart: moved shapes are answered by this refit or by a rebuild.
 * Triangle vertices (bvhgpu_tree_set_triangles) are the caller's to refresh. */
int bvhgpu_refit_f32(bvhgpu_tree *tree, const float *aabbs, size_t n, int mem);
int bvhgpu_refit_f64(bvhgpu_tree *tree, const double *aabbs, size_t n, int mem);
void bvhgpu_tree_destroy(bvhgpu_tree *tree);

/* ---- asynchronous step.  The calls above return when their result is complete (one host round trip each).  These
 * enqueue the same work on the ctx's stream and return at once, so that ONE host thread can keep several steps in flight
 * (on several ctxs = several streams: the latency-bound build of one step overlaps the traversal of another).
 *   bvhgpu_rebuild_flat_async_*  = bvhgpu_rebuild_flat_* (Bvh::build_par + Bvh::flatten) without the wait.  `aabbs` must stay
 *                                  valid until the wait (BVHGPU_HOST: pinned memory, or the copy is synchronous anyway).
 *   bvhgpu_traverse_async_*      = bvhgpu_traverse_* on rays resident in HBM (mem must be BVHGPU_DEVICE); the tree may
 *                                  still be building on the same stream.
 *   bvhgpu_hits_wait             completes the batch: waits for the stream, completes the tree's build (input validation;
 *                                  an unbalanced tree is finished level by level) and replays the batch if the optimistic
 *                                  launch was not enough (hit pool too small, tree not finished when the walk ran).  The
 *                                  statuses the synchronous calls would have returned are returned here.
 *   bvhgpu_tree_wait             the same for a tree alone.  Every other entry point that looks at a tree waits by itself.
 * Lifetimes: the rays of an asynchronous batch must stay valid until its bvhgpu_hits_wait (a replay reads them again).  The tree
 * may be waited for, traversed again, rebuilt or destroyed in any order: every result object remembers which generation of the tree
 * it walked and whether that generation was still unfinalized, so its wait replays exactly when needed, whoever finalized the
 * build first; a rebuild / refit / import / destroy of the tree first completes the batches still in flight on it (their own
 * bvhgpu_hits_wait then returns what that completion found).  bvhgpu_hits_fetch* / _info / _device return BVHGPU_INVALID_ARG on a
 * result object whose batch has not been waited for. */
int bvhgpu_rebuild_flat_async_f32(bvhgpu_tree *tree, const float *aabbs, size_t n, int mem);
int bvhgpu_rebuild_flat_async_f64(bvhgpu_tree *tree, const double *aabbs, size_t n, int mem);
int bvhgpu_tree_wait(bvhgpu_tree *tree);

int bvhgpu_tree_info(const bvhgpu_tree *tree, int *dtype, size_t *n_shapes, size_t *n_nodes, size_t *n_flat);
/* Vec<BvhNode> (bvh_impl.rs:27-33): 2n-1 entries of bvhgpu_node_f32/_f64. */
int bvhgpu_tree_nodes(bvhgpu_tree *tree, void *out, int mem);
/* the argument of BHShape::set_bh_node_index for every shape (bvh_node.rs:102; bounding_hierarchy.rs:53-65). */
int bvhgpu_tree_shape_nodes(bvhgpu_tree *tree, uint32_t *out, int mem);
/* number of level-synchronous passes the last build needed (diagnostic). */
int bvhgpu_tree_build_levels(const bvhgpu_tree *tree, int *levels);

/* ---- flatten: replaces Bvh::flatten / flatten_custom (flat_bvh.rs:240-251, 312-319).  Produces the
 * reference-layout FlatNode array (3n-2 entries) and the engine's own traversal array in HBM. ---- */
int bvhgpu_flatten(bvhgpu_tree *tree);
int bvhgpu_flat_nodes(bvhgpu_tree *tree, void *out, int mem);
/* Upload a FlatBvh built elsewhere (e.g. by the Rust crate through flatten_custom with a #[repr(C)]
 * constructor) together with the shapes' current AABBs; replaces FlatBvh ownership (flat_bvh.rs:254). */
int bvhgpu_tree_from_flat_f32(bvhgpu_ctx *ctx, const bvhgpu_flat_f32 *flat, size_t n_flat, const float *shape_aabbs,
                              size_t n, bvhgpu_tree **out);
int bvhgpu_tree_from_flat_f64(bvhgpu_ctx *ctx, const bvhgpu_flat_f64 *flat, size_t n_flat, const double *shape_aabbs,
                              size_t n, bvhgpu_tree **out);

/* ---- scene transport for multi-GPU: one contiguous blob (traversal array + shape AABBs + LDS slot
 * table + triangle vertices when set) that the
 * caller broadcasts (RCCL via torch.distributed / ncclBroadcast) and imports on the peers.  A tree
 * imported this way supports traversal only.  For bvhgpu_scene_import *out may point to NULL (a
 * tree is allocated) or to a tree a previous bvhgpu_scene_import returned (its HBM is reused). ---- */
int bvhgpu_scene_nbytes(const bvhgpu_tree *tree, size_t *nbytes);
int bvhgpu_scene_export(bvhgpu_tree *tree, void *dst, int mem);
int bvhgpu_scene_import(bvhgpu_ctx *ctx, const void *src, size_t nbytes, int mem, bvhgpu_tree **out);

/* ---- multi-GPU: rays shard across the GPUs of a node, the tree is read-only after flatten — the path has
 * exactly ONE exchange step, a broadcast of the root's flattened tree to the peers, done here with RCCL over xGMI
 * straight out of / into the trees' own HBM buffers (traversal array, shape AABBs, LDS slot table, optionally the
 * triangle vertices).  Hit lists stay on the GPU that produced them.  The reference has no counterpart (single
 * address space); this is what replaces sharing `&FlatBvh` between rayon workers.
 * A communicator is formed either by ONE process over `ndev` ctxs (one per GPU, ncclCommInitAll), or by one process per
 * GPU: rank 0 calls bvhgpu_comm_unique_id, the launcher carries the 128 bytes to the peers (MPI, a torch.distributed
 * store, a file …) and every rank calls bvhgpu_comm_init_rank.  `trees` has one entry per LOCAL device of the
 * communicator (one entry in the process-per-GPU form); the root's entry is the source, a peer's entry is NULL (a tree
 * is allocated) or the result of an earlier broadcast / scene import on that ctx (its HBM is reused).  Received trees
 * support traversal and point queries only (no BvhNode array).  The calls are collective: every rank must make them. */
typedef struct bvhgpu_comm bvhgpu_comm;
#define BVHGPU_COMM_ID_BYTES 128
#define BVHGPU_BCAST_TRIANGLES 1u
int bvhgpu_comm_unique_id(void *id_out /* BVHGPU_COMM_ID_BYTES */);
int bvhgpu_comm_init_rank(bvhgpu_ctx *ctx, int nranks, int rank, const void *id, bvhgpu_comm **out);
int bvhgpu_comm_init_all(bvhgpu_ctx *const *ctxs, int ndev, bvhgpu_comm **out);
int bvhgpu_comm_info(const bvhgpu_comm *comm, int *nranks, int *first_rank, int *n_local);
void bvhgpu_comm_destroy(bvhgpu_comm *comm);
/* Which RCCL the broadcasts go through: `version` = ncclGetVersion's code (major*10000 + minor*100 + patch; 0 if the library
 * has no such entry point), `shared_with_process` = 1 when the copy was already loaded in the process (PyTorch bundles one) and
 * no second RCCL was opened, `library_path` = the file the entry points were resolved from (truncated to cap).  Loads RCCL on
 * first use like every bvhgpu_comm_* call; BVHGPU_RCCL_ERROR when there is none.  Launchers print it so that a scaling record
 * shows how many ranks over which library it ran (bench.py's "rccl" object). */
int bvhgpu_rccl_info(int *version, int *shared_with_process, char *library_path, size_t cap);
/* peers learn type and size from a 64-byte header that travels first (one host round trip per rank).  If the root has no valid
 * tree (NULL, not flattened, invalid input) it still sends the header: its own call returns the reason, the peers' calls return
 * BVHGPU_INVALID_ARG — nobody is left waiting inside a collective. */
int bvhgpu_bcast(bvhgpu_comm *comm, bvhgpu_tree **trees, int root);
/* every rank already knows dtype and shape count (a frame loop over a scene of constant size): one group of broadcasts
 * (status header + arrays) enqueued on the ctxs' streams, no host round trip on any rank; `what`: BVHGPU_BCAST_TRIANGLES to send
 * the vertices too.  The root's tree must be a tree built (or received) here with exactly that dtype / shape count; it may
 * still be building (bvhgpu_rebuild_flat_async_*): the header is then composed on the device from the build's own status.
 * The peers' trees are usable on their streams at once (bvhgpu_traverse_async_*); the header is looked at when a peer's tree
 * is first waited for: BVHGPU_INVALID_ARG = the root had nothing valid to send (the root's call or wait returned the reason),
 * BVHGPU_REBROADCAST = see the status codes.  A received tree carries the root's `exact_only` property (a split without SAH
 * winner, bvh_node.rs:225-230: no wide walk), as does a scene blob. */
int bvhgpu_bcast_known(bvhgpu_comm *comm, bvhgpu_tree **trees, int root, int dtype, size_t n_shapes, unsigned what);

/* ---- rays ---- */
/* Ray::new (ray_impl.rs:70-80) for a batch: normalise, cache 1/d.  origins/dirs: n x 3. */
int bvhgpu_rays_new_f32(bvhgpu_ctx *ctx, const float *origins, const float *dirs, size_t n, int mem_in,
                        bvhgpu_ray_f32 *out, int mem_out);
int bvhgpu_rays_new_f64(bvhgpu_ctx *ctx, const double *origins, const double *dirs, size_t n, int mem_in,
                        bvhgpu_ray_f64 *out, int mem_out);
/* The bench harness' ray stream create_ray(seed, bounds) (testbase.rs:687-691, seed 0 at :825),
 * rays [first, first+n) generated directly in HBM (`out` is device memory). */
int bvhgpu_gen_rays_f32(bvhgpu_ctx *ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f32 *out_dev);
/* same stream widened to f64 AFTER generation (f32 origin/target → f64 Ray::new), for the f64 config. */
int bvhgpu_gen_rays_f64(bvhgpu_ctx *ctx, uint64_t first, size_t n, const float bounds[6], bvhgpu_ray_f64 *out_dev);

/* ---- point query: replaces <FlatBvh as BoundingHierarchy>::nearest_to (flat_bvh.rs:513-562; trait
 * bounding_hierarchy.rs:262-336) for n query points (n x 3).  The shape's PointDistance::distance_squared is a user
 * callback in the reference; the two the reference's harness defines are built in: kind 0 = the shape's own
 * Aabb::min_distance_squared (UnitBox, testbase.rs:101-105; aabb_impl.rs:618-629), kind 1 = closest point on the
 * triangle (testbase.rs:367-443, needs bvhgpu_tree_set_triangles).  out_shape[i] = shape index (BVHGPU_NONE for an
 * empty hierarchy), out_dist[i] = the distance (not squared, :561).  `mem` applies to all three buffers. ---- */
int bvhgpu_nearest_f32(bvhgpu_tree *tree, const float *points, size_t n, int mem, int kind, uint32_t *out_shape, float *out_dist);
int bvhgpu_nearest_f64(bvhgpu_tree *tree, const double *points, size_t n, int mem, int kind, uint32_t *out_shape, double *out_dist);

/* Ray::intersects_triangle (ray_impl.rs:154-213) for n independent pairs: ray i against triangle i
 * (tris: n x [a xyz, b xyz, c xyz]); out: n x {distance,u,v}.  `mem` applies to all three buffers. */
int bvhgpu_ray_triangle_pairs_f32(bvhgpu_ctx *ctx, const bvhgpu_ray_f32 *rays, const float *tris, size_t n, int mem, float *out);
int bvhgpu_ray_triangle_pairs_f64(bvhgpu_ctx *ctx, const bvhgpu_ray_f64 *rays, const double *tris, size_t n, int mem, double *out);

/* Coherent primary rays (BASELINE.json configs[2]; the reference has no camera, this is the engine's definition,
 * which the tests' CPU checker restates operation by operation).  cam = eye[3], right[3], up[3], forward[3], tan_x, tan_y.
 * Ray `first + i` belongs to pixel x = id % width, y = id / width:  sx = (((x + 0.5) / W) * 2) - 1,
 * sy = 1 - (((y + 0.5) / H) * 2),  dir = (forward + (sx * tan_x) * right) + (sy * tan_y) * up  (separately
 * rounded f32 operations), ray = Ray::new(eye, dir) (ray_impl.rs:70-80).  `out_dev` is device memory. */
int bvhgpu_gen_primary_rays_f32(bvhgpu_ctx *ctx, const float cam[14], uint32_t width, uint32_t height, uint64_t first,
                                size_t n, bvhgpu_ray_f32 *out_dev);
int bvhgpu_gen_primary_rays_f64(bvhgpu_ctx *ctx, const float cam[14], uint32_t width, uint32_t height, uint64_t first,
                                size_t n, bvhgpu_ray_f64 *out_dev);

/* ---- traverse: replaces <FlatBvh as BoundingHierarchy>::traverse (flat_bvh.rs:396-431) and, by the
 * equivalence of bvh_node.rs:288-319, Bvh::traverse (bvh_impl.rs:104-119), for a BATCH of rays.
 * Result = CSR: offsets[n_rays+1], indices[total]; ray i's shapes are indices[offsets[i]..offsets[i+1])
 * in the reference's order (flat-array / DFS left-first order).
 * *hits may point to NULL (a result object is allocated) or to a previous result (buffers reused). ---- */
int bvhgpu_traverse_f32(bvhgpu_tree *tree, const bvhgpu_ray_f32 *rays, size_t n_rays, int mem, unsigned flags,
                        bvhgpu_hits **hits);
int bvhgpu_traverse_f64(bvhgpu_tree *tree, const bvhgpu_ray_f64 *rays, size_t n_rays, int mem, unsigned flags,
                        bvhgpu_hits **hits);

/* ---- the same for a caller whose rays live in HOST memory and who wants the hit lists back in host memory (ABI 7): what
 * GpuBvh::traverse_batch of the Rust shim does, in one call.  Replaces `for ray in rays { flat.traverse(&Ray::new(o, d), shapes) }`
 * (ray_impl.rs:70-80, flat_bvh.rs:396-431).
 *   origins, directions   n_rays x 3 T each: Ray::new runs on the device (normalised direction and 1/d correctly rounded: the bits
 *                         Ray::new gives) — 24 bytes per ray cross the link instead of the 36 of a Ray struct.
 *                         directions == NULL: `origins` points to n_rays Ray structs (bvhgpu_ray_f32 / _f64), used as they are.
 *   offsets               n_rays + 1 entries, always written.
 *   indices, indices_cap  written when the batch's hit total fits (total <= indices_cap); otherwise the call still succeeds,
 *                         *total says what is needed and bvhgpu_traverse_host_indices fetches them (no second traversal).
 *   flags                 0, BVHGPU_TRAVERSE_COHERENT, BVHGPU_TRAVERSE_RAYS_OD6 (origins = n_rays x [o xyz, d xyz], directions ignored).
 * The tree may still be building (bvhgpu_rebuild_flat_async_* with BVHGPU_HOST shapes): the ray upload does not wait for the build.
 * The batch is walked in chunks (BVHGPU_TUNE_HOST_CHUNKS: by default three, the last one an eighth of the batch) on three streams — upload of
 * chunk k+1, Ray::new + walk of chunk k, download of the offsets of chunk k-1 — with one host wait at the end; with pinned buffers (bvhgpu_host_alloc / _register) the copies are DMA at link
 * speed.  Results are those of bvhgpu_traverse_* on the same rays, byte for byte.  Returns when everything is in the caller's buffers. */
int bvhgpu_traverse_host_f32(bvhgpu_tree *tree, const float *origins, const float *directions, size_t n_rays, unsigned flags,
                             uint32_t *offsets, uint32_t *indices, size_t indices_cap, uint64_t *total);
int bvhgpu_traverse_host_f64(bvhgpu_tree *tree, const double *origins, const double *directions, size_t n_rays, unsigned flags,
                             uint32_t *offsets, uint32_t *indices, size_t indices_cap, uint64_t *total);
int bvhgpu_traverse_host_indices(bvhgpu_ctx *ctx, uint32_t *indices, size_t indices_cap);
/* GpuBvh::build + traverse_batch of a frame in ONE call: bvhgpu_rebuild_flat_async_*(aabbs, BVHGPU_HOST) + bvhgpu_traverse_host_*, with the
 * ray upload enqueued FIRST — it is the long pole (24 MB against the build's 3) and depends on nothing — so that the build (Bvh::build_par +
 * flatten, bvh_impl.rs:40-96, flat_bvh.rs:240-319) runs underneath it.  Same results, same errors (NaN / inf shapes: BVHGPU_INVALID_ARG,
 * nothing usable built). */
int bvhgpu_build_traverse_host_f32(bvhgpu_tree *tree, const float *aabbs, size_t n, const float *origins, const float *directions, size_t n_rays,
                                   unsigned flags, uint32_t *offsets, uint32_t *indices, size_t indices_cap, uint64_t *total);
int bvhgpu_build_traverse_host_f64(bvhgpu_tree *tree, const double *aabbs, size_t n, const double *origins, const double *directions, size_t n_rays,
                                   unsigned flags, uint32_t *offsets, uint32_t *indices, size_t indices_cap, uint64_t *total);
int bvhgpu_traverse_async_f32(bvhgpu_tree *tree, const bvhgpu_ray_f32 *rays, size_t n_rays, int mem, unsigned flags,
                              bvhgpu_hits **hits);
int bvhgpu_traverse_async_f64(bvhgpu_tree *tree, const bvhgpu_ray_f64 *rays, size_t n_rays, int mem, unsigned flags,
                              bvhgpu_hits **hits);
int bvhgpu_hits_wait(bvhgpu_hits *hits);
/* Triangle vertices of the shapes (n x [a xyz, b xyz, c xyz], the fields of testbase.rs Triangle :316-323) for the
 * TRIANGLES / CLOSEST flags; n must equal the tree's shape count.  The tree keeps its own HBM copy. */
int bvhgpu_tree_set_triangles_f32(bvhgpu_tree *tree, const float *verts, size_t n, int mem);
int bvhgpu_tree_set_triangles_f64(bvhgpu_tree *tree, const double *verts, size_t n, int mem);
int bvhgpu_hits_info(const bvhgpu_hits *hits, size_t *n_rays, uint64_t *total, bvhgpu_traverse_stats *stats);
/* Which form of the walk produced the batch the result object holds (diagnostic; the lists never depend on it): */
#define BVHGPU_WALK_WIDE 1u       /* the 4-wide walk (batches of BVHGPU_TUNE_TRAVERSE_LDS_MIN_RAYS rays and more) */
#define BVHGPU_WALK_STAGED 2u     /* ... with a ray's first hits handed over through its own slot (BVHGPU_TUNE_WIDE_STAGE_SHIFT) */
#define BVHGPU_WALK_REC8 4u       /* ... with pool records of 8 bytes per hit (BVHGPU_TUNE_WIDE_REC8) */
#define BVHGPU_WALK_F64_GUIDE 8u  /* ... over the f32 guide boxes of an f64 tree, leaf candidates confirmed in f64 (BVHGPU_TUNE_WIDE_F64_GUIDE) */
int bvhgpu_hits_walk_info(const bvhgpu_hits *hits, unsigned *flags);
/* ... and the name of the walk kernel it was handed to, NUL-terminated, spelled the way rocprofv3 prints kernel names (e.g.
 * "bvhgpu::k_traverse_wide<float, 0, 2, 1024, 8, 0>"): what a profile of the same call must be looked up under (diagnostic, ABI 6) */
int bvhgpu_hits_walk_kernel(const bvhgpu_hits *hits, char *name, size_t cap);
/* copy out; indices / tslice may be NULL.  tslice: 2 scalars of the tree's dtype per hit (flag T_SLICE). */
int bvhgpu_hits_fetch(bvhgpu_hits *hits, uint32_t *offsets, uint32_t *indices, void *tslice, int mem);
/* TRIANGLES: 3 scalars {distance,u,v} per hit, CSR order (distance = +inf: no intersection, ray_impl.rs:150-151). */
int bvhgpu_hits_fetch_triangles(bvhgpu_hits *hits, void *isect, int mem);
/* CLOSEST: per ray {distance,u,v} (+inf,0,0 when nothing is hit) and the shape index (BVHGPU_NONE when nothing is hit). */
int bvhgpu_hits_fetch_closest(bvhgpu_hits *hits, void *isect, uint32_t *shape, int mem);
/* borrow the device arrays (valid until the next traverse into / destroy of this result). */
int bvhgpu_hits_device(const bvhgpu_hits *hits, const uint32_t **offsets, const uint32_t **indices, const void **tslice);
void bvhgpu_hits_destroy(bvhgpu_hits *hits);

/* ---- timing hook used by bench.py: HIP-event time (ms) of the kernels of the last call of each
 * phase on this ctx's stream (build / flatten / traverse main kernel / traverse total). ---- */
typedef struct { float build_ms, flatten_ms, traverse_kernel_ms, traverse_total_ms; } bvhgpu_timings;
int bvhgpu_enable_timing(bvhgpu_ctx *ctx, int on);
int bvhgpu_last_timings(bvhgpu_ctx *ctx, bvhgpu_timings *out);

/* ---- scene ingest (host code): Wavefront OBJ → triangles, as `obj::load_obj::<Triangle>` + `FromRawVertex::process`
 * do for the reference's Sponza benches (testbase.rs:445-487, 619-634): every polygon (P / PT / PN / PTN) becomes a
 * triangle fan over its positions.  *tris_out: malloc'ed n x 9 floats [a xyz, b xyz, c xyz] (release with
 * bvhgpu_obj_free); bounds_out (nullable): join of the triangle AABBs.  Errors: BVHGPU_INVALID_ARG + bvhgpu_obj_last_error(). */
int bvhgpu_obj_parse(const char *text, size_t len, float **tris_out, size_t *n_tris_out, float bounds_out[6]);
void bvhgpu_obj_free(float *tris);
const char *bvhgpu_obj_last_error(void);
/* Triangle::new's cached aabb = empty.grow(a).grow(b).grow(c) (testbase.rs:325-333): n x [min xyz, max xyz] */
int bvhgpu_triangles_aabbs_f32(const float *tris, size_t n, float *aabbs_out);

/* ---- tuning knobs (performance only; results never change).  Not part of the reference surface. ---- */
typedef enum {
    BVHGPU_TUNE_TRAVERSE_VARIANT = 0,      /* 0 one ray per lane per launch; 2 persistent workgroups over the binary traversal
                                              array with ray refill and the top of the tree resident in LDS; 3 (default) wide
                                              walk (four grandchild boxes per step) where its preconditions hold, else 2 */
    BVHGPU_TUNE_WIDE_ITEMS_LOG4 = 1,       /* variant 3, CSR outputs: cut every ray into up to 4^v items (v = 0, 1, 2); default -1 = by batch size */
    BVHGPU_TUNE_WIDE_STACK_LDS = 2,        /* variant 3: stack entries per lane kept in LDS (default -1 = 6 for rays cut into items, 8 for whole rays, 10 for whole rays of a COHERENT batch; deeper entries live in HBM) */
    BVHGPU_TUNE_TRAVERSE_LDS_MIN_RAYS = 3, /* variant 2 is used for batches of at least this many rays (default 16384) */
    BVHGPU_TUNE_TRAVERSE_LDS_SLOTS = 4,    /* variant 2: top-of-tree entries kept in LDS per workgroup (default 0 = as many as let two workgroups share a CU: f32 2559, f64 1462) */
    BVHGPU_TUNE_TRAVERSE_LDS_THREADS = 5,  /* variant 2: workgroup size (default 0 = per type: f32 1024, f64 512) */
    BVHGPU_TUNE_TRAVERSE_SPLIT = 6,        /* variant 2, CSR outputs: walk every ray as two items (left / right subtree of the
                                              root); default 1 */
    BVHGPU_TUNE_WIDE_WG_PER_CU = 7,        /* variant 3: workgroups sharing a CU's LDS (default 0 = 2) */
    BVHGPU_TUNE_WIDE_THREADS = 8,          /* variant 3: workgroup size (default 0 = per type: f32 1024, f64 512) */
    BVHGPU_TUNE_WIDE_SLOTS = 9,            /* variant 3: cap on the top-of-tree wide nodes kept in LDS (default 0 = what fits) */
    BVHGPU_TUNE_BUILD_LEVEL_LAUNCHES = 10, /* builder, level tier: 1 = one launch per tree level (k_level: split of level L-1 and binning of level L fused,
                                              the selection recomputed per tile: the shorter chain, more work per shape); 2 = two launches per level
                                              (k_bin, k_split); 0 (default) = by scene size: 1 up to 250 000 shapes, 2 above */
    BVHGPU_TUNE_WIDE_EARLY_ITEMS = 11,     /* variant 3 with BVHGPU_TRAVERSE_RAYS_READY on a tree that is being rebuilt: 1 = cut the rays into items on a
                                              side stream as soon as the build has split tree level 3; 0 (default) = in the walk's prologue.  Measured on
                                              configs[1]: the walk shrinks by 10 µs, the two stream dependencies and the guest kernel cost the build as
                                              much — kept selectable, off by default */
    BVHGPU_TUNE_WIDE_STAGE_SHIFT = 12,     /* variant 3, whole rays (large batches), indices only: the first 2^v shapes of every ray are written straight to a
                                              per-ray slot (4 bytes per hit) and gathered into the CSR; only later hits of a ray go through 12-byte pool
                                              records.  -1 (default) = 3 for batches flagged BVHGPU_TRAVERSE_COHERENT, off otherwise; 0 = off (every hit
                                              through the pool); 2 .. 5 = on for every such batch */
    BVHGPU_TUNE_WIDE_REC8 = 13,            /* variant 3, whole rays, indices only: pool records of 8 bytes per hit (16-byte records {ray, k, shape, shape} for two
                                              consecutive hits of a ray) instead of 12; 1 (default) on, 0 off */
    BVHGPU_TUNE_WIDE_F64_GUIDE = 14,       /* variant 3, f64 trees, indices only: 1 (default) = walk the tree's f32 guide boxes (the f64 boxes grown by 2^-18 of the
                                              scene's largest |coordinate| and rounded outward) with the rays converted to f32 where the walk loads them, and test only the leaf
                                              candidates in f64 — the hit lists are the same, the walk runs at the f32 rate; a batch with a ray outside
                                              the range the argument covers (|origin| > 3 x scene, |1/d| x scene outside 2^+-100, non-finite) is replayed
                                              with the f64 walk and the result object skips the guide for 1, 2, 4 … 64 batches on consecutive failures before it tries again; 0 = always the f64 walk */
    BVHGPU_TUNE_FLATTEN_LAZY = 15,         /* bvhgpu_rebuild_flat_async / bvhgpu_build_flat_*: 1 (default) = the flatten behind a build writes what the wide walk
                                              reads (wide nodes, their LDS slot table, an f64 tree's guide nodes); the reference-layout FlatNode array and the
                                              folded binary array are written by a second pass the first time something asks for them (bvhgpu_flat_nodes,
                                              a binary / STATS / t-slice / ordered walk, nearest_to, scene export, a broadcast) — same arrays, byte for byte;
                                              0 = every flatten writes everything at once; 2 = every flatten writes everything, the second pass on the ctx's side
                                              stream BESIDE the walk that follows (the walk reads none of it); the batch's wait covers it; 3 = every flatten writes
                                              the reference-layout FlatNode array (what Bvh::flatten returns) and the wide walk's arrays; only the engine's own folded
                                              binary array waits for its first reader */
    BVHGPU_TUNE_BUILD_LEVEL_PERSIST = 16,  /* builder, level tier with one launch per level, scenes of 32 x 769 shapes and more: != 0 = the tier's passes from tree level 4 on run as
                                              ONE persistent launch, a workgroup group per level-3 subtree that synchronises with itself after every pass (1 = 32 workgroups
                                              per group, 8 .. 64 = that many); 0 (default) = a launch per level throughout.  Measured on configs[1]: 64 per group builds in
                                              0.1886 - 0.1891 ms against 0.1876 - 0.1895 ms (the pass is a chain of dependent loads, not its launch boundary), 32 per group is
                                              slower (two tiles per workgroup) — kept selectable and parity-tested, off by default (profiles/r5_persist_ab.log) */
    BVHGPU_TUNE_HOST_CHUNKS = 17,          /* bvhgpu_traverse_host_*: the batch is walked as this many chunks, so that the upload of one overlaps the walk of the
                                              previous one and the download of the offsets of the one before (1 .. 16); 0 (default) = by batch size: 3 from 512 K
                                              rays, 2 from 256 K, else 1.  The last chunk is an eighth of the batch (what is left to do when the upload ends) */
    BVHGPU_TUNE_WIDE_MIN_RAYS_PER_WG = 18, /* variant 3, rays cut into 16 items, batches that fill at most a quarter of the chip's workgroup slots at one ray per lane
                                              (up to 128 K rays): the rays are spread over all slots, down to this many rays per workgroup (multiple of 64; default
                                              256); 0 = one ray per lane always */
    BVHGPU_TUNE_HOST_ZERO_COPY = 19,       /* bvhgpu_traverse_host_* / bvhgpu_build_traverse_host_* on PINNED buffers (bvhgpu_host_alloc / _register): bit 1 = the device
                                              writes offsets / indices straight into the caller's arrays (no download, no hop to a third stream); bit 0 = the device
                                              reads the caller's ray arrays itself (Ray::new straight out of host memory, no staging copy).  Default 2: bit 0 moves the
                                              bytes at link speed (56 GB/s) but the build running beside it slows down four-fold while a kernel streams host memory
                                              (k_level 12 -> 50 µs: profiles/r6_host_zero_copy.log), so the copy engines keep the upload.  0 = copy engines both
                                              ways.  Pageable buffers always go through the copy engines */
    BVHGPU_TUNE_BUILD_LEVEL_TILE = 20,     /* builder, level tier, two launches per level (scenes above 250 K shapes): positions per workgroup tile, a multiple of 256
                                              (0, default: 512, from 600 K shapes 1024, from 2 M 2048, from 6 M 4096).  A scheduling unit only: the tree is the same whatever the tile */
    BVHGPU_TUNE_FLATTEN_INLINE = 21,       /* f32 trees, the flatten enqueued with a build: the builder's wave tier writes the FlatNode / wide-node entries of the
                                              subtrees it builds (<= 64 shapes: 97 % of the nodes) itself and the flatten kernel behind it only the nodes above:
                                              1 (default), or 0: the flatten kernel writes everything.  Same arrays either way */
    BVHGPU_TUNE_COUNT = 22
} bvhgpu_tune;
int bvhgpu_set_tuning(bvhgpu_ctx *ctx, int knob, int value);
int bvhgpu_get_tuning(const bvhgpu_ctx *ctx, int knob, int *value);

#ifdef __cplusplus
}
#endif
#endif /* BVH_MI355X_H */
