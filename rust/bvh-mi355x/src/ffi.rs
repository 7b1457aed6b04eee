//! Declarations of include/bvh_mi355x.h (ABI version 7), one for one.  Every function returns a `bvhgpu_status`
//! (0 = OK) and never unwinds; `bvhgpu_last_error` gives the text of the last failure on a ctx.
#![allow(non_camel_case_types, dead_code)]
use core::ffi::{c_char, c_int, c_uint, c_void};

pub const BVHGPU_ABI_VERSION: c_int = 7;
pub const BVHGPU_NONE: u32 = u32::MAX; // flat_bvh.rs:51-53

// bvhgpu_status
pub const BVHGPU_OK: c_int = 0;
pub const BVHGPU_INVALID_ARG: c_int = 1;
pub const BVHGPU_HIP_ERROR: c_int = 2;
pub const BVHGPU_OOM: c_int = 3;
pub const BVHGPU_OVERFLOW: c_int = 4;
pub const BVHGPU_NO_DEVICE: c_int = 5;
pub const BVHGPU_DTYPE_MISMATCH: c_int = 6;
pub const BVHGPU_NOT_FLATTENED: c_int = 7;
pub const BVHGPU_RCCL_ERROR: c_int = 8;
pub const BVHGPU_REBROADCAST: c_int = 9;
// bvhgpu_dtype / bvhgpu_mem
pub const BVHGPU_F32: c_int = 0;
pub const BVHGPU_F64: c_int = 1;
pub const BVHGPU_HOST: c_int = 0;
pub const BVHGPU_DEVICE: c_int = 1;
// traversal flags
pub const BVHGPU_TRAVERSE_T_SLICE: c_uint = 1;
pub const BVHGPU_TRAVERSE_STATS: c_uint = 2;
pub const BVHGPU_TRAVERSE_TRIANGLES: c_uint = 4;
pub const BVHGPU_TRAVERSE_CLOSEST: c_uint = 8;
pub const BVHGPU_TRAVERSE_COHERENT: c_uint = 16;
pub const BVHGPU_TRAVERSE_NEAREST_FIRST: c_uint = 32;
pub const BVHGPU_TRAVERSE_FARTHEST_FIRST: c_uint = 64;
pub const BVHGPU_TRAVERSE_BEST_FIRST: c_uint = 128;
pub const BVHGPU_TRAVERSE_RAYS_READY: c_uint = 256;
pub const BVHGPU_TRAVERSE_RAYS_OD6: c_uint = 512;
// bvhgpu_hits_walk_info
pub const BVHGPU_WALK_WIDE: c_uint = 1;
pub const BVHGPU_WALK_STAGED: c_uint = 2;
pub const BVHGPU_WALK_REC8: c_uint = 4;
pub const BVHGPU_WALK_F64_GUIDE: c_uint = 8;
pub const BVHGPU_COMM_ID_BYTES: usize = 128;
pub const BVHGPU_BCAST_TRIANGLES: c_uint = 1;

#[repr(C)] pub struct bvhgpu_ctx { _p: [u8; 0] }
#[repr(C)] pub struct bvhgpu_tree { _p: [u8; 0] }
#[repr(C)] pub struct bvhgpu_hits { _p: [u8; 0] }
#[repr(C)] pub struct bvhgpu_comm { _p: [u8; 0] }

/// POD image of `enum BvhNode<f32,3>` (src/bvh/bvh_node.rs:21-47); `shape == u32::MAX` ⇒ inner node
#[repr(C)] #[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct bvhgpu_node_f32 {
    pub l_min: [f32; 3], pub l_max: [f32; 3], pub r_min: [f32; 3], pub r_max: [f32; 3],
    pub parent: u32, pub l: u32, pub r: u32, pub shape: u32,
}
#[repr(C)] #[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct bvhgpu_node_f64 {
    pub l_min: [f64; 3], pub l_max: [f64; 3], pub r_min: [f64; 3], pub r_max: [f64; 3],
    pub parent: u32, pub l: u32, pub r: u32, pub shape: u32,
}
/// same field order as `struct FlatNode<f32,3>` (src/flat_bvh.rs:17-46)
#[repr(C)] #[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct bvhgpu_flat_f32 { pub min: [f32; 3], pub max: [f32; 3], pub entry: u32, pub exit: u32, pub shape: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct bvhgpu_flat_f64 { pub min: [f64; 3], pub max: [f64; 3], pub entry: u32, pub exit: u32, pub shape: u32, pub _pad: u32 }
/// same field order as `struct Ray<f32,3>` (src/ray/ray_impl.rs:17-29)
#[repr(C)] #[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct bvhgpu_ray_f32 { pub o: [f32; 3], pub d: [f32; 3], pub inv: [f32; 3] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct bvhgpu_ray_f64 { pub o: [f64; 3], pub d: [f64; 3], pub inv: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct bvhgpu_traverse_stats { pub hits: u64, pub visited: u64, pub leaf_visits: u64, pub device_steps: u64, pub wave_steps: u64 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct bvhgpu_timings { pub build_ms: f32, pub flatten_ms: f32, pub traverse_kernel_ms: f32, pub traverse_total_ms: f32 }

const _: () = assert!(core::mem::size_of::<bvhgpu_node_f32>() == 64 && core::mem::size_of::<bvhgpu_node_f64>() == 112);
const _: () = assert!(core::mem::size_of::<bvhgpu_flat_f32>() == 36 && core::mem::size_of::<bvhgpu_flat_f64>() == 64);
const _: () = assert!(core::mem::size_of::<bvhgpu_ray_f32>() == 36 && core::mem::size_of::<bvhgpu_ray_f64>() == 72);

extern "C" {
    // context
    pub fn bvhgpu_abi_version() -> c_int;
    pub fn bvhgpu_device_count(out: *mut c_int) -> c_int;
    pub fn bvhgpu_status_string(status: c_int) -> *const c_char;
    pub fn bvhgpu_create(device: c_int, stream: *mut c_void, out: *mut *mut bvhgpu_ctx) -> c_int;
    pub fn bvhgpu_destroy(ctx: *mut bvhgpu_ctx);
    pub fn bvhgpu_last_error(ctx: *const bvhgpu_ctx) -> *const c_char;
    pub fn bvhgpu_synchronize(ctx: *mut bvhgpu_ctx) -> c_int;
    pub fn bvhgpu_stream(ctx: *mut bvhgpu_ctx) -> *mut c_void;
    pub fn bvhgpu_device_alloc(ctx: *mut bvhgpu_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn bvhgpu_device_free(ctx: *mut bvhgpu_ctx, ptr: *mut c_void) -> c_int;
    pub fn bvhgpu_device_copy(ctx: *mut bvhgpu_ctx, dst: *mut c_void, dst_mem: c_int, src: *const c_void, src_mem: c_int, bytes: usize) -> c_int;
    pub fn bvhgpu_host_alloc(ctx: *mut bvhgpu_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn bvhgpu_host_free(ctx: *mut bvhgpu_ctx, ptr: *mut c_void) -> c_int;
    pub fn bvhgpu_host_register(ctx: *mut bvhgpu_ctx, ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn bvhgpu_host_unregister(ctx: *mut bvhgpu_ctx, ptr: *mut c_void) -> c_int;
    // build: Bvh::build / build_par / build_with_executor (bvh_impl.rs:40-96)
    pub fn bvhgpu_build_f32(ctx: *mut bvhgpu_ctx, aabbs: *const f32, n: usize, mem: c_int, out: *mut *mut bvhgpu_tree) -> c_int;
    pub fn bvhgpu_build_f64(ctx: *mut bvhgpu_ctx, aabbs: *const f64, n: usize, mem: c_int, out: *mut *mut bvhgpu_tree) -> c_int;
    pub fn bvhgpu_rebuild_f32(t: *mut bvhgpu_tree, aabbs: *const f32, n: usize, mem: c_int) -> c_int;
    pub fn bvhgpu_rebuild_f64(t: *mut bvhgpu_tree, aabbs: *const f64, n: usize, mem: c_int) -> c_int;
    // FlatBvh::build (flat_bvh.rs:328-331)
    pub fn bvhgpu_build_flat_f32(ctx: *mut bvhgpu_ctx, aabbs: *const f32, n: usize, mem: c_int, out: *mut *mut bvhgpu_tree) -> c_int;
    pub fn bvhgpu_build_flat_f64(ctx: *mut bvhgpu_ctx, aabbs: *const f64, n: usize, mem: c_int, out: *mut *mut bvhgpu_tree) -> c_int;
    pub fn bvhgpu_rebuild_flat_f32(t: *mut bvhgpu_tree, aabbs: *const f32, n: usize, mem: c_int) -> c_int;
    pub fn bvhgpu_rebuild_flat_f64(t: *mut bvhgpu_tree, aabbs: *const f64, n: usize, mem: c_int) -> c_int;
    // fix_aabbs_ascending over the whole tree (optimization.rs:355-391)
    pub fn bvhgpu_refit_f32(t: *mut bvhgpu_tree, aabbs: *const f32, n: usize, mem: c_int) -> c_int;
    pub fn bvhgpu_refit_f64(t: *mut bvhgpu_tree, aabbs: *const f64, n: usize, mem: c_int) -> c_int;
    // asynchronous step
    pub fn bvhgpu_rebuild_flat_async_f32(t: *mut bvhgpu_tree, aabbs: *const f32, n: usize, mem: c_int) -> c_int;
    pub fn bvhgpu_rebuild_flat_async_f64(t: *mut bvhgpu_tree, aabbs: *const f64, n: usize, mem: c_int) -> c_int;
    pub fn bvhgpu_tree_wait(t: *mut bvhgpu_tree) -> c_int;
    pub fn bvhgpu_tree_destroy(t: *mut bvhgpu_tree);
    pub fn bvhgpu_tree_info(t: *const bvhgpu_tree, dtype: *mut c_int, n: *mut usize, n_nodes: *mut usize, n_flat: *mut usize) -> c_int;
    pub fn bvhgpu_tree_nodes(t: *mut bvhgpu_tree, out: *mut c_void, mem: c_int) -> c_int; // Vec<BvhNode>
    pub fn bvhgpu_tree_shape_nodes(t: *mut bvhgpu_tree, out: *mut u32, mem: c_int) -> c_int; // set_bh_node_index arguments
    pub fn bvhgpu_tree_build_levels(t: *const bvhgpu_tree, levels: *mut c_int) -> c_int;
    // flatten: Bvh::flatten / flatten_custom (flat_bvh.rs:240-251, 312-319)
    pub fn bvhgpu_flatten(t: *mut bvhgpu_tree) -> c_int;
    pub fn bvhgpu_flat_nodes(t: *mut bvhgpu_tree, out: *mut c_void, mem: c_int) -> c_int;
    pub fn bvhgpu_tree_from_flat_f32(ctx: *mut bvhgpu_ctx, flat: *const bvhgpu_flat_f32, n_flat: usize, shape_aabbs: *const f32, n: usize, out: *mut *mut bvhgpu_tree) -> c_int;
    pub fn bvhgpu_tree_from_flat_f64(ctx: *mut bvhgpu_ctx, flat: *const bvhgpu_flat_f64, n_flat: usize, shape_aabbs: *const f64, n: usize, out: *mut *mut bvhgpu_tree) -> c_int;
    // scene transport (blob) and the RCCL exchange step
    pub fn bvhgpu_scene_nbytes(t: *const bvhgpu_tree, nbytes: *mut usize) -> c_int;
    pub fn bvhgpu_scene_export(t: *mut bvhgpu_tree, dst: *mut c_void, mem: c_int) -> c_int;
    pub fn bvhgpu_scene_import(ctx: *mut bvhgpu_ctx, src: *const c_void, nbytes: usize, mem: c_int, out: *mut *mut bvhgpu_tree) -> c_int;
    pub fn bvhgpu_comm_unique_id(id_out: *mut c_void) -> c_int;
    pub fn bvhgpu_comm_init_rank(ctx: *mut bvhgpu_ctx, nranks: c_int, rank: c_int, id: *const c_void, out: *mut *mut bvhgpu_comm) -> c_int;
    pub fn bvhgpu_comm_init_all(ctxs: *const *mut bvhgpu_ctx, ndev: c_int, out: *mut *mut bvhgpu_comm) -> c_int;
    pub fn bvhgpu_comm_info(comm: *const bvhgpu_comm, nranks: *mut c_int, first_rank: *mut c_int, n_local: *mut c_int) -> c_int;
    pub fn bvhgpu_comm_destroy(comm: *mut bvhgpu_comm);
    pub fn bvhgpu_rccl_info(version: *mut c_int, shared_with_process: *mut c_int, library_path: *mut c_char, cap: usize) -> c_int;
    pub fn bvhgpu_bcast(comm: *mut bvhgpu_comm, trees: *mut *mut bvhgpu_tree, root: c_int) -> c_int;
    pub fn bvhgpu_bcast_known(comm: *mut bvhgpu_comm, trees: *mut *mut bvhgpu_tree, root: c_int, dtype: c_int, n_shapes: usize, what: c_uint) -> c_int;
    // rays: Ray::new (ray_impl.rs:70-80), the bench stream (testbase.rs:687-691), primary rays
    pub fn bvhgpu_rays_new_f32(ctx: *mut bvhgpu_ctx, origins: *const f32, dirs: *const f32, n: usize, mem_in: c_int, out: *mut bvhgpu_ray_f32, mem_out: c_int) -> c_int;
    pub fn bvhgpu_rays_new_f64(ctx: *mut bvhgpu_ctx, origins: *const f64, dirs: *const f64, n: usize, mem_in: c_int, out: *mut bvhgpu_ray_f64, mem_out: c_int) -> c_int;
    pub fn bvhgpu_gen_rays_f32(ctx: *mut bvhgpu_ctx, first: u64, n: usize, bounds: *const f32, out_dev: *mut bvhgpu_ray_f32) -> c_int;
    pub fn bvhgpu_gen_rays_f64(ctx: *mut bvhgpu_ctx, first: u64, n: usize, bounds: *const f32, out_dev: *mut bvhgpu_ray_f64) -> c_int;
    pub fn bvhgpu_gen_primary_rays_f32(ctx: *mut bvhgpu_ctx, cam: *const f32, width: u32, height: u32, first: u64, n: usize, out_dev: *mut bvhgpu_ray_f32) -> c_int;
    pub fn bvhgpu_gen_primary_rays_f64(ctx: *mut bvhgpu_ctx, cam: *const f32, width: u32, height: u32, first: u64, n: usize, out_dev: *mut bvhgpu_ray_f64) -> c_int;
    // nearest_to (flat_bvh.rs:513-562) and Ray::intersects_triangle pairs (ray_impl.rs:154-213)
    pub fn bvhgpu_nearest_f32(t: *mut bvhgpu_tree, points: *const f32, n: usize, mem: c_int, kind: c_int, out_shape: *mut u32, out_dist: *mut f32) -> c_int;
    pub fn bvhgpu_nearest_f64(t: *mut bvhgpu_tree, points: *const f64, n: usize, mem: c_int, kind: c_int, out_shape: *mut u32, out_dist: *mut f64) -> c_int;
    pub fn bvhgpu_ray_triangle_pairs_f32(ctx: *mut bvhgpu_ctx, rays: *const bvhgpu_ray_f32, tris: *const f32, n: usize, mem: c_int, out: *mut f32) -> c_int;
    pub fn bvhgpu_ray_triangle_pairs_f64(ctx: *mut bvhgpu_ctx, rays: *const bvhgpu_ray_f64, tris: *const f64, n: usize, mem: c_int, out: *mut f64) -> c_int;
    // traverse: FlatBvh::traverse (flat_bvh.rs:396-431) for a batch → CSR
    pub fn bvhgpu_traverse_f32(t: *mut bvhgpu_tree, rays: *const bvhgpu_ray_f32, n_rays: usize, mem: c_int, flags: c_uint, hits: *mut *mut bvhgpu_hits) -> c_int;
    pub fn bvhgpu_traverse_f64(t: *mut bvhgpu_tree, rays: *const bvhgpu_ray_f64, n_rays: usize, mem: c_int, flags: c_uint, hits: *mut *mut bvhgpu_hits) -> c_int;
    pub fn bvhgpu_traverse_host_f32(t: *mut bvhgpu_tree, origins: *const f32, directions: *const f32, n_rays: usize, flags: c_uint, offsets: *mut u32, indices: *mut u32, indices_cap: usize, total: *mut u64) -> c_int;
    pub fn bvhgpu_traverse_host_f64(t: *mut bvhgpu_tree, origins: *const f64, directions: *const f64, n_rays: usize, flags: c_uint, offsets: *mut u32, indices: *mut u32, indices_cap: usize, total: *mut u64) -> c_int;
    pub fn bvhgpu_traverse_host_indices(ctx: *mut bvhgpu_ctx, indices: *mut u32, indices_cap: usize) -> c_int;
    pub fn bvhgpu_build_traverse_host_f32(t: *mut bvhgpu_tree, aabbs: *const f32, n: usize, origins: *const f32, directions: *const f32, n_rays: usize, flags: c_uint, offsets: *mut u32, indices: *mut u32, indices_cap: usize, total: *mut u64) -> c_int;
    pub fn bvhgpu_build_traverse_host_f64(t: *mut bvhgpu_tree, aabbs: *const f64, n: usize, origins: *const f64, directions: *const f64, n_rays: usize, flags: c_uint, offsets: *mut u32, indices: *mut u32, indices_cap: usize, total: *mut u64) -> c_int;
    pub fn bvhgpu_traverse_async_f32(t: *mut bvhgpu_tree, rays: *const bvhgpu_ray_f32, n_rays: usize, mem: c_int, flags: c_uint, hits: *mut *mut bvhgpu_hits) -> c_int;
    pub fn bvhgpu_traverse_async_f64(t: *mut bvhgpu_tree, rays: *const bvhgpu_ray_f64, n_rays: usize, mem: c_int, flags: c_uint, hits: *mut *mut bvhgpu_hits) -> c_int;
    pub fn bvhgpu_hits_wait(h: *mut bvhgpu_hits) -> c_int;
    pub fn bvhgpu_tree_set_triangles_f32(t: *mut bvhgpu_tree, verts: *const f32, n: usize, mem: c_int) -> c_int;
    pub fn bvhgpu_tree_set_triangles_f64(t: *mut bvhgpu_tree, verts: *const f64, n: usize, mem: c_int) -> c_int;
    pub fn bvhgpu_hits_info(h: *const bvhgpu_hits, n_rays: *mut usize, total: *mut u64, stats: *mut bvhgpu_traverse_stats) -> c_int;
    pub fn bvhgpu_hits_walk_info(h: *const bvhgpu_hits, flags: *mut c_uint) -> c_int;
    pub fn bvhgpu_hits_walk_kernel(h: *const bvhgpu_hits, name: *mut c_char, cap: usize) -> c_int;
    pub fn bvhgpu_hits_fetch(h: *mut bvhgpu_hits, offsets: *mut u32, indices: *mut u32, tslice: *mut c_void, mem: c_int) -> c_int;
    pub fn bvhgpu_hits_fetch_triangles(h: *mut bvhgpu_hits, isect: *mut c_void, mem: c_int) -> c_int;
    pub fn bvhgpu_hits_fetch_closest(h: *mut bvhgpu_hits, isect: *mut c_void, shape: *mut u32, mem: c_int) -> c_int;
    pub fn bvhgpu_hits_device(h: *const bvhgpu_hits, offsets: *mut *const u32, indices: *mut *const u32, tslice: *mut *const c_void) -> c_int;
    pub fn bvhgpu_hits_destroy(h: *mut bvhgpu_hits);
    // timing, scene ingest, tuning
    pub fn bvhgpu_enable_timing(ctx: *mut bvhgpu_ctx, on: c_int) -> c_int;
    pub fn bvhgpu_last_timings(ctx: *mut bvhgpu_ctx, out: *mut bvhgpu_timings) -> c_int;
    pub fn bvhgpu_obj_parse(text: *const c_char, len: usize, tris_out: *mut *mut f32, n_tris_out: *mut usize, bounds_out: *mut f32) -> c_int;
    pub fn bvhgpu_obj_free(tris: *mut f32);
    pub fn bvhgpu_obj_last_error() -> *const c_char;
    pub fn bvhgpu_triangles_aabbs_f32(tris: *const f32, n: usize, aabbs_out: *mut f32) -> c_int;
    pub fn bvhgpu_set_tuning(ctx: *mut bvhgpu_ctx, knob: c_int, value: c_int) -> c_int;
    pub fn bvhgpu_get_tuning(ctx: *const bvhgpu_ctx, knob: c_int, value: *mut c_int) -> c_int;
}
