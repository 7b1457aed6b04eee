//! `GpuBvh<T>`: the `bvh` crate's `BoundingHierarchy<T, 3>` (src/bounding_hierarchy.rs:89-336) on an MI355X, for `T = f32`
//! and `T = f64`, over the C ABI of libbvh_mi355x.so (include/bvh_mi355x.h).  Build / flatten / batched traversal run on the
//! GPU and give the arrays the crate's own `Bvh::build` + `Bvh::flatten` + `FlatBvh::traverse` give, bit for bit (see
//! DESIGN.md §2); queries that need user callbacks (`IntersectsAabb` for anything but rays, arbitrary `PointDistance`) run the
//! crate's own loops over the downloaded flat array.
//!
//! The scalar type is a sealed trait (`GpuScalar`, implemented for `f32` and `f64` only — the two instantiations the
//! engine has): it names the `#[repr(C)]` images of `BvhNode` / `FlatNode` / `Ray` for that type and the `_f32` / `_f64` entry
//! points, so that `GpuBvh<T>` is written once.  `GpuBvh32` / `GpuBvh64` are the two aliases.
//!
//! Which GPU: `GpuBvh::from_aabbs(aabbs, device)` takes it explicitly; the trait's `build` (whose signature the crate fixes
//! and has no room for it) uses `default_device()`, a process-wide setting (`set_default_device`; initial value 0, or
//! `BVH_MI355X_DEVICE` from the environment).  Every `GpuBvh` remembers its device (`device()`).
//!
//! NOTE: written against bvh 0.12.0 / nalgebra 0.34 by reading their sources; the image this engine was developed in has
//! no cargo/rustc, so this crate has NOT been compiled there.  tests/test_abi_cpu.py checks every `extern "C"` declaration
//! of ffi.rs against the header (names, argument counts, argument and return types) on every CPU run.
#![allow(clippy::missing_safety_doc)]
pub mod ffi;

use bvh::aabb::{Aabb, IntersectsAabb};
use bvh::bounding_hierarchy::{BHShape, BHValue, BoundingHierarchy};
use bvh::bvh::{Bvh, BvhNode, BvhNodeBuildArgs};
use bvh::flat_bvh::{FlatBvh, FlatNode};
use bvh::point_query::PointDistance;
use bvh::ray::{Intersection, Ray};
use core::ffi::{c_int, c_uint, c_void};
use core::sync::atomic::{AtomicI32, Ordering};
use nalgebra::{Point3, Vector3};

/// Panics with the engine's message: the crate has no error type, contract violations panic there too
/// (e.g. NaN centroids, src/bvh/bvh_node.rs:214-217).
fn check(ctx: *const ffi::bvhgpu_ctx, rc: c_int) {
    if rc != ffi::BVHGPU_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(ffi::bvhgpu_last_error(ctx)) };
        panic!("bvh_mi355x: status {rc}: {}", msg.to_string_lossy());
    }
}

static DEFAULT_DEVICE: AtomicI32 = AtomicI32::new(-1);

/// The GPU the trait's `build` / `build_par` use (their signatures are the crate's and carry no device argument).
pub fn default_device() -> i32 {
    let d = DEFAULT_DEVICE.load(Ordering::Relaxed);
    if d >= 0 {
        return d;
    }
    std::env::var("BVH_MI355X_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0)
}
/// One process per GPU: call this once with the local rank before the first `build`.
pub fn set_default_device(device: i32) {
    DEFAULT_DEVICE.store(device, Ordering::Relaxed);
}
/// Number of MI355X devices the engine sees (0 ⇒ every call fails with `BVHGPU_NO_DEVICE`: there is no CPU fallback)
pub fn device_count() -> i32 {
    let mut n: c_int = 0;
    unsafe { ffi::bvhgpu_device_count(&mut n) };
    n
}

mod sealed {
    pub trait Sealed {}
    impl Sealed for f32 {}
    impl Sealed for f64 {}
}

/// The two scalar types the engine is instantiated for.  Everything type-dependent at the boundary lives here.
pub trait GpuScalar: BHValue + sealed::Sealed + Default + 'static {
    /// `#[repr(C)]` image of `enum BvhNode<T,3>` (src/bvh/bvh_node.rs:21-47)
    type Node: Copy + Default;
    /// `#[repr(C)]` image of `struct FlatNode<T,3>` (src/flat_bvh.rs:17-46)
    type Flat: Copy + Default;
    /// `#[repr(C)]` image of `struct Ray<T,3>` (src/ray/ray_impl.rs:17-29)
    type RayC: Copy + Default;
    const DTYPE: c_int;

    unsafe fn build_flat(ctx: *mut ffi::bvhgpu_ctx, aabbs: *const Self, n: usize, mem: c_int, out: *mut *mut ffi::bvhgpu_tree) -> c_int;
    unsafe fn rebuild_flat(t: *mut ffi::bvhgpu_tree, aabbs: *const Self, n: usize, mem: c_int) -> c_int;
    unsafe fn rebuild_flat_async(t: *mut ffi::bvhgpu_tree, aabbs: *const Self, n: usize, mem: c_int) -> c_int;
    #[allow(clippy::too_many_arguments)]
    unsafe fn traverse_host(t: *mut ffi::bvhgpu_tree, origins: *const Self, directions: *const Self, n: usize, flags: c_uint, offsets: *mut u32, indices: *mut u32, cap: usize, total: *mut u64) -> c_int;
    #[allow(clippy::too_many_arguments)]
    unsafe fn build_traverse_host(t: *mut ffi::bvhgpu_tree, aabbs: *const Self, n_shapes: usize, origins: *const Self, directions: *const Self, n: usize, flags: c_uint, offsets: *mut u32, indices: *mut u32, cap: usize, total: *mut u64) -> c_int;
    unsafe fn refit(t: *mut ffi::bvhgpu_tree, aabbs: *const Self, n: usize, mem: c_int) -> c_int;
    unsafe fn traverse(t: *mut ffi::bvhgpu_tree, rays: *const Self::RayC, n: usize, mem: c_int, flags: c_uint, hits: *mut *mut ffi::bvhgpu_hits) -> c_int;
    unsafe fn set_triangles(t: *mut ffi::bvhgpu_tree, verts: *const Self, n: usize, mem: c_int) -> c_int;
    unsafe fn tree_from_flat(ctx: *mut ffi::bvhgpu_ctx, flat: *const Self::Flat, n_flat: usize, shape_aabbs: *const Self, n: usize, out: *mut *mut ffi::bvhgpu_tree) -> c_int;

    fn node_to_crate(raw: &Self::Node) -> BvhNode<Self, 3>;
    fn flat_to_crate(raw: &Self::Flat) -> FlatNode<Self, 3>;
    fn flat_from_parts(aabb: &Aabb<Self, 3>, entry: u32, exit: u32, shape: u32) -> Self::Flat;
    fn ray_to_ffi(r: &Ray<Self, 3>) -> Self::RayC;
}

macro_rules! impl_gpu_scalar {
    ($t:ty, $dtype:expr, $node:ident, $flat:ident, $ray:ident, $build_flat:ident, $rebuild_flat:ident, $refit:ident, $traverse:ident,
     $set_tris:ident, $from_flat:ident, $rebuild_async:ident, $traverse_host:ident, $build_traverse_host:ident, $flat_ctor:expr) => {
        impl GpuScalar for $t {
            type Node = ffi::$node;
            type Flat = ffi::$flat;
            type RayC = ffi::$ray;
            const DTYPE: c_int = $dtype;
            unsafe fn build_flat(ctx: *mut ffi::bvhgpu_ctx, aabbs: *const $t, n: usize, mem: c_int, out: *mut *mut ffi::bvhgpu_tree) -> c_int {
                ffi::$build_flat(ctx, aabbs, n, mem, out)
            }
            unsafe fn rebuild_flat(t: *mut ffi::bvhgpu_tree, aabbs: *const $t, n: usize, mem: c_int) -> c_int {
                ffi::$rebuild_flat(t, aabbs, n, mem)
            }
            unsafe fn rebuild_flat_async(t: *mut ffi::bvhgpu_tree, aabbs: *const $t, n: usize, mem: c_int) -> c_int {
                ffi::$rebuild_async(t, aabbs, n, mem)
            }
            unsafe fn traverse_host(t: *mut ffi::bvhgpu_tree, origins: *const $t, directions: *const $t, n: usize, flags: c_uint, offsets: *mut u32, indices: *mut u32, cap: usize, total: *mut u64) -> c_int {
                ffi::$traverse_host(t, origins, directions, n, flags, offsets, indices, cap, total)
            }
            unsafe fn build_traverse_host(t: *mut ffi::bvhgpu_tree, aabbs: *const $t, n_shapes: usize, origins: *const $t, directions: *const $t, n: usize, flags: c_uint, offsets: *mut u32, indices: *mut u32, cap: usize, total: *mut u64) -> c_int {
                ffi::$build_traverse_host(t, aabbs, n_shapes, origins, directions, n, flags, offsets, indices, cap, total)
            }
            unsafe fn refit(t: *mut ffi::bvhgpu_tree, aabbs: *const $t, n: usize, mem: c_int) -> c_int {
                ffi::$refit(t, aabbs, n, mem)
            }
            unsafe fn traverse(t: *mut ffi::bvhgpu_tree, rays: *const ffi::$ray, n: usize, mem: c_int, flags: c_uint, hits: *mut *mut ffi::bvhgpu_hits) -> c_int {
                ffi::$traverse(t, rays, n, mem, flags, hits)
            }
            unsafe fn set_triangles(t: *mut ffi::bvhgpu_tree, verts: *const $t, n: usize, mem: c_int) -> c_int {
                ffi::$set_tris(t, verts, n, mem)
            }
            unsafe fn tree_from_flat(ctx: *mut ffi::bvhgpu_ctx, flat: *const ffi::$flat, n_flat: usize, shape_aabbs: *const $t, n: usize, out: *mut *mut ffi::bvhgpu_tree) -> c_int {
                ffi::$from_flat(ctx, flat, n_flat, shape_aabbs, n, out)
            }
            fn node_to_crate(r: &ffi::$node) -> BvhNode<$t, 3> {
                if r.shape != ffi::BVHGPU_NONE {
                    BvhNode::Leaf { parent_index: r.parent as usize, shape_index: r.shape as usize }
                } else {
                    BvhNode::Node {
                        parent_index: r.parent as usize,
                        child_l_index: r.l as usize,
                        child_l_aabb: Aabb::with_bounds(Point3::from(r.l_min), Point3::from(r.l_max)),
                        child_r_index: r.r as usize,
                        child_r_aabb: Aabb::with_bounds(Point3::from(r.r_min), Point3::from(r.r_max)),
                    }
                }
            }
            fn flat_to_crate(f: &ffi::$flat) -> FlatNode<$t, 3> {
                FlatNode {
                    aabb: Aabb::with_bounds(Point3::from(f.min), Point3::from(f.max)),
                    entry_index: f.entry,
                    exit_index: f.exit,
                    shape_index: f.shape,
                }
            }
            fn flat_from_parts(aabb: &Aabb<$t, 3>, entry: u32, exit: u32, shape: u32) -> ffi::$flat {
                let ctor: fn([$t; 3], [$t; 3], u32, u32, u32) -> ffi::$flat = $flat_ctor;
                ctor([aabb.min.x, aabb.min.y, aabb.min.z], [aabb.max.x, aabb.max.y, aabb.max.z], entry, exit, shape)
            }
            fn ray_to_ffi(r: &Ray<$t, 3>) -> ffi::$ray {
                // Ray { origin, direction, inv_direction } (src/ray/ray_impl.rs:17-29): copied field by field, never transmuted
                ffi::$ray {
                    o: [r.origin.x, r.origin.y, r.origin.z],
                    d: [r.direction.x, r.direction.y, r.direction.z],
                    inv: [r.inv_direction.x, r.inv_direction.y, r.inv_direction.z],
                }
            }
        }
    };
}
impl_gpu_scalar!(f32, ffi::BVHGPU_F32, bvhgpu_node_f32, bvhgpu_flat_f32, bvhgpu_ray_f32, bvhgpu_build_flat_f32, bvhgpu_rebuild_flat_f32,
                 bvhgpu_refit_f32, bvhgpu_traverse_f32, bvhgpu_tree_set_triangles_f32, bvhgpu_tree_from_flat_f32,
                 bvhgpu_rebuild_flat_async_f32, bvhgpu_traverse_host_f32, bvhgpu_build_traverse_host_f32,
                 |min, max, entry, exit, shape| ffi::bvhgpu_flat_f32 { min, max, entry, exit, shape });
impl_gpu_scalar!(f64, ffi::BVHGPU_F64, bvhgpu_node_f64, bvhgpu_flat_f64, bvhgpu_ray_f64, bvhgpu_build_flat_f64, bvhgpu_rebuild_flat_f64,
                 bvhgpu_refit_f64, bvhgpu_traverse_f64, bvhgpu_tree_set_triangles_f64, bvhgpu_tree_from_flat_f64,
                 bvhgpu_rebuild_flat_async_f64, bvhgpu_traverse_host_f64, bvhgpu_build_traverse_host_f64,
                 |min, max, entry, exit, shape| ffi::bvhgpu_flat_f64 { min, max, entry, exit, shape, _pad: 0 });

fn aabb_to_6<T: GpuScalar>(b: &Aabb<T, 3>) -> [T; 6] {
    [b.min.x, b.min.y, b.min.z, b.max.x, b.max.y, b.max.z]
}

pub fn ray_to_ffi<T: GpuScalar>(r: &Ray<T, 3>) -> T::RayC {
    T::ray_to_ffi(r)
}

pub struct GpuBvh<T: GpuScalar> {
    ctx: *mut ffi::bvhgpu_ctx,
    tree: *mut ffi::bvhgpu_tree,
    device: i32,
    n_shapes: usize,
    /// CPU copy of the flat array in the crate's own layout, for the generic queries of the trait
    flat: FlatBvh<T, 3>,
    /// `rebuild_async` ran and `sync_flat` has not: the CPU copy is the previous tree's
    flat_stale: bool,
}
/// `BoundingHierarchy<f32, 3>` on the GPU (BASELINE configs[1]-[3])
pub type GpuBvh32 = GpuBvh<f32>;
/// `BoundingHierarchy<f64, 3>` on the GPU (BASELINE configs[4]: tree, rays and every test that decides a hit in double precision)
pub type GpuBvh64 = GpuBvh<f64>;

// the handles are only used through &self / &mut self; the engine's ctx is not internally locked: one GpuBvh per thread
unsafe impl<T: GpuScalar> Send for GpuBvh<T> {}

/// CSR result of a batch: ray i hit `indices[offsets[i]..offsets[i+1]]`, in the order `FlatBvh::traverse` returns them
pub struct BatchHits {
    pub offsets: Vec<u32>,
    pub indices: Vec<u32>,
}

/// Result of the reference harness' inner loop for one ray (src/testbase.rs:826-836): the nearest
/// `Ray::intersects_triangle` over the candidates `FlatBvh::traverse` returns; `shape == u32::MAX` and
/// `distance == +inf` when the ray hits nothing
pub struct ClosestHit<T> {
    pub intersection: Intersection<T>,
    pub shape: u32,
}

/// A fixed-length slice in pinned host memory (`bvhgpu_host_alloc` / `bvhgpu_host_free`); derefs to `[U]`.  Must not outlive the
/// `GpuBvh` it came from.
pub struct PinnedVec<U> {
    ctx: *mut ffi::bvhgpu_ctx,
    ptr: *mut U,
    len: usize,
}
impl<U> core::ops::Deref for PinnedVec<U> {
    type Target = [U];
    fn deref(&self) -> &[U] {
        unsafe { core::slice::from_raw_parts(self.ptr, self.len) }
    }
}
impl<U> core::ops::DerefMut for PinnedVec<U> {
    fn deref_mut(&mut self) -> &mut [U] {
        unsafe { core::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}
impl<U> Drop for PinnedVec<U> {
    fn drop(&mut self) {
        unsafe { ffi::bvhgpu_host_free(self.ctx, self.ptr.cast()) };
    }
}

impl<T: GpuScalar> GpuBvh<T> {
    /// Bvh::build_par + Bvh::flatten on GPU `device` from the shapes' AABBs (n x [min xyz, max xyz])
    pub fn from_aabbs(aabbs: &[[T; 6]], device: i32) -> GpuBvh<T> {
        let mut ctx = core::ptr::null_mut();
        let mut tree = core::ptr::null_mut();
        unsafe {
            check(ctx, ffi::bvhgpu_create(device, core::ptr::null_mut(), &mut ctx));
            check(ctx, T::build_flat(ctx, aabbs.as_ptr().cast(), aabbs.len(), ffi::BVHGPU_HOST, &mut tree));
        }
        let mut me = GpuBvh { ctx, tree, device, n_shapes: aabbs.len(), flat: Vec::new(), flat_stale: false };
        me.flat = me.download_flat();
        me
    }

    /// the GPU this hierarchy lives on
    pub fn device(&self) -> i32 {
        self.device
    }

    /// the argument of `BHShape::set_bh_node_index` for every shape (src/bvh/bvh_node.rs:102)
    pub fn shape_nodes(&self) -> Vec<u32> {
        let mut sn = vec![0u32; self.n_shapes];
        unsafe { check(self.ctx, ffi::bvhgpu_tree_shape_nodes(self.tree, sn.as_mut_ptr(), ffi::BVHGPU_HOST)); }
        sn
    }

    /// `Vec<BvhNode>` exactly as `Bvh::build` produces it (bit-identical AABBs, same indices)
    pub fn to_bvh(&self) -> Bvh<T, 3> {
        let nn = if self.n_shapes == 0 { 0 } else { 2 * self.n_shapes - 1 };
        let mut raw = vec![T::Node::default(); nn];
        unsafe { check(self.ctx, ffi::bvhgpu_tree_nodes(self.tree, raw.as_mut_ptr().cast(), ffi::BVHGPU_HOST)); }
        Bvh { nodes: raw.iter().map(T::node_to_crate).collect() }
    }

    fn download_flat(&self) -> FlatBvh<T, 3> {
        let nf = if self.n_shapes >= 2 { 3 * self.n_shapes - 2 } else { self.n_shapes };
        let mut raw = vec![T::Flat::default(); nf];
        unsafe { check(self.ctx, ffi::bvhgpu_flat_nodes(self.tree, raw.as_mut_ptr().cast(), ffi::BVHGPU_HOST)); }
        raw.iter().map(T::flat_to_crate).collect()
    }

    /// `FlatBvh::traverse` (src/flat_bvh.rs:396-431) for many rays at once — what the GPU is for
    pub fn traverse_batch(&self, rays: &[Ray<T, 3>]) -> BatchHits {
        // `bvhgpu_traverse_host_*` with directions == NULL: the crate's own Ray structs (origin, direction, inv_direction), uploaded in
        // chunks beside the walk of the previous chunk, CSR offsets downloaded behind it — one call, one host wait
        let r: Vec<T::RayC> = rays.iter().map(T::ray_to_ffi).collect();
        self.host_batch(r.as_ptr().cast(), core::ptr::null(), r.len())
    }

    /// The same for rays given as origins and directions (what `Ray::new` takes, src/ray/ray_impl.rs:70-80): `Ray::new` runs on the
    /// device — the correctly rounded divide and square root give its bits — and 24 bytes per ray cross the link instead of 36.
    /// With both slices in pinned memory (`PinnedVec`) the copies are DMA at link speed.
    pub fn traverse_batch_od(&self, origins: &[[T; 3]], directions: &[[T; 3]]) -> BatchHits {
        assert_eq!(origins.len(), directions.len(), "one direction per origin");
        self.host_batch(origins.as_ptr().cast(), directions.as_ptr().cast(), origins.len())
    }

    /// One frame of a host-resident caller in one call: `rebuild(aabbs)` + `traverse_batch_od(origins, directions)`, the ray upload
    /// enqueued before the build so that the build runs underneath it (`bvhgpu_build_traverse_host_*`).  The CPU copy the trait's
    /// generic queries walk is refreshed by `sync_flat()`.
    pub fn rebuild_and_traverse(&mut self, aabbs: &[[T; 6]], origins: &[[T; 3]], directions: &[[T; 3]]) -> BatchHits {
        assert_eq!(origins.len(), directions.len(), "one direction per origin");
        let n = origins.len();
        let mut offsets = vec![0u32; n + 1];
        let mut indices = vec![0u32; n.max(1 << 16)];
        let mut total = 0u64;
        unsafe {
            check(self.ctx, T::build_traverse_host(self.tree, aabbs.as_ptr().cast(), aabbs.len(), origins.as_ptr().cast(), directions.as_ptr().cast(), n, 0,
                                                   offsets.as_mut_ptr(), indices.as_mut_ptr(), indices.len(), &mut total));
            if total as usize > indices.len() {
                indices.resize(total as usize, 0);
                check(self.ctx, ffi::bvhgpu_traverse_host_indices(self.ctx, indices.as_mut_ptr(), indices.len()));
            }
        }
        self.n_shapes = aabbs.len();
        self.flat_stale = true;
        indices.truncate(total as usize);
        BatchHits { offsets, indices }
    }

    /// ... and as ONE slice of `[o.x, o.y, o.z, d.x, d.y, d.z]` per ray (`BVHGPU_TRAVERSE_RAYS_OD6`): one transfer per chunk instead of two
    pub fn traverse_batch_od6(&self, rays: &[[T; 6]]) -> BatchHits {
        self.host_batch_flags(rays.as_ptr().cast(), core::ptr::null(), rays.len(), ffi::BVHGPU_TRAVERSE_RAYS_OD6)
    }

    fn host_batch(&self, origins: *const T, directions: *const T, n: usize) -> BatchHits {
        self.host_batch_flags(origins, directions, n, 0)
    }

    fn host_batch_flags(&self, origins: *const T, directions: *const T, n: usize, flags: c_uint) -> BatchHits {
        let mut offsets = vec![0u32; n + 1];
        let mut indices = vec![0u32; n.max(1 << 16)];
        let mut total = 0u64;
        unsafe {
            check(self.ctx, T::traverse_host(self.tree, origins, directions, n, flags, offsets.as_mut_ptr(), indices.as_mut_ptr(), indices.len(), &mut total));
            if total as usize > indices.len() {
                indices.resize(total as usize, 0);
                check(self.ctx, ffi::bvhgpu_traverse_host_indices(self.ctx, indices.as_mut_ptr(), indices.len()));
            }
        }
        indices.truncate(total as usize);
        BatchHits { offsets, indices }
    }

    /// `rebuild` without the wait: the build is enqueued and the call returns; a `traverse_batch*` that follows uploads its rays
    /// beside it.  `aabbs` must stay untouched until the next call that looks at the tree (pinned memory: the copy is asynchronous).
    pub fn rebuild_async(&mut self, aabbs: &[[T; 6]]) {
        unsafe { check(self.ctx, T::rebuild_flat_async(self.tree, aabbs.as_ptr().cast(), aabbs.len(), ffi::BVHGPU_HOST)); }
        self.n_shapes = aabbs.len();
        self.flat_stale = true;
    }

    /// Pinned host memory for `n` elements on this hierarchy's ctx (`bvhgpu_host_alloc`): read and written by the DMA engines directly
    pub fn pinned<U: Copy + Default>(&self, n: usize) -> PinnedVec<U> {
        let mut p = core::ptr::null_mut();
        unsafe { check(self.ctx, ffi::bvhgpu_host_alloc(self.ctx, n * core::mem::size_of::<U>(), &mut p)); }
        let v = PinnedVec { ctx: self.ctx, ptr: p.cast::<U>(), len: n };
        for i in 0..n {
            unsafe { v.ptr.add(i).write(U::default()) };
        }
        v
    }

    /// The triangle stage needs the vertices (one triangle per shape, n x [a xyz, b xyz, c xyz]): src/testbase.rs:325-333
    pub fn set_triangles(&mut self, verts: &[[T; 9]]) {
        assert_eq!(verts.len(), self.n_shapes, "one triangle per shape");
        unsafe { check(self.ctx, T::set_triangles(self.tree, verts.as_ptr().cast(), verts.len(), ffi::BVHGPU_HOST)); }
    }

    /// The harness loop of the reference's benches for a whole batch (src/testbase.rs:826-836): per ray, `FlatBvh::traverse` then
    /// `Ray::intersects_triangle` (src/ray/ray_impl.rs:154-213) on every candidate, keeping the nearest — fused into the walk on
    /// the GPU (`BVHGPU_TRAVERSE_CLOSEST`).  Needs `set_triangles`.
    pub fn traverse_closest(&self, rays: &[Ray<T, 3>]) -> Vec<ClosestHit<T>> {
        let r: Vec<T::RayC> = rays.iter().map(T::ray_to_ffi).collect();
        let mut hits = core::ptr::null_mut();
        let mut isect = vec![[T::default(); 3]; rays.len()];
        let mut shape = vec![0u32; rays.len()];
        unsafe {
            check(self.ctx, T::traverse(self.tree, r.as_ptr(), r.len(), ffi::BVHGPU_HOST, ffi::BVHGPU_TRAVERSE_CLOSEST, &mut hits));
            check(self.ctx, ffi::bvhgpu_hits_fetch_closest(hits, isect.as_mut_ptr() as *mut c_void, shape.as_mut_ptr(), ffi::BVHGPU_HOST));
            ffi::bvhgpu_hits_destroy(hits);
        }
        isect.iter().zip(shape).map(|(i, s)| ClosestHit { intersection: Intersection::new(i[0], i[1], i[2]), shape: s }).collect()
    }

    /// Build again from new AABBs into the same device buffers (a frame loop: no allocation in the steady state)
    pub fn rebuild(&mut self, aabbs: &[[T; 6]]) {
        unsafe { check(self.ctx, T::rebuild_flat(self.tree, aabbs.as_ptr().cast(), aabbs.len(), ffi::BVHGPU_HOST)); }
        self.n_shapes = aabbs.len();
        self.flat = self.download_flat();
        self.flat_stale = false;
    }

    /// After `rebuild_async`: wait for the build and refresh the CPU copy the trait's generic queries walk
    pub fn sync_flat(&mut self) {
        self.flat = self.download_flat();
        self.flat_stale = false;
    }

    /// the shapes moved, the topology stays: `Bvh::fix_aabbs_ascending` (src/bvh/optimization.rs:355-391) over the whole tree
    pub fn refit(&mut self, aabbs: &[[T; 6]]) {
        unsafe { check(self.ctx, T::refit(self.tree, aabbs.as_ptr().cast(), aabbs.len(), ffi::BVHGPU_HOST)); }
        self.flat = self.download_flat();
    }

    pub fn raw(&self) -> (*mut ffi::bvhgpu_ctx, *mut ffi::bvhgpu_tree) {
        (self.ctx, self.tree)
    }
}

impl<T: GpuScalar> BoundingHierarchy<T, 3> for GpuBvh<T> {
    fn build<Shape: BHShape<T, 3>>(shapes: &mut [Shape]) -> GpuBvh<T> {
        // (1) the only callback the device cannot make: shape.aabb(), gathered once per shape
        let aabbs: Vec<[T; 6]> = shapes.iter().map(|s| aabb_to_6(&s.aabb())).collect();
        // (2) Bvh::build_par + Bvh::flatten on the GPU the process chose (set_default_device)
        let bh = GpuBvh::from_aabbs(&aabbs, default_device());
        // (3) set_bh_node_index for every shape (src/bvh/bvh_node.rs:102)
        for (s, ni) in shapes.iter_mut().zip(bh.shape_nodes()) {
            s.set_bh_node_index(ni as usize);
        }
        bh
    }

    fn build_with_executor<
        Shape: BHShape<T, 3>,
        Executor: FnMut(BvhNodeBuildArgs<'_, Shape, T, 3>, BvhNodeBuildArgs<'_, Shape, T, 3>),
    >(
        shapes: &mut [Shape],
        _executor: Executor,
    ) -> GpuBvh<T> {
        // the executor only schedules CPU sub-builds and cannot change the result (node placement is arithmetic,
        // src/bvh/bvh_node.rs:138-142): the GPU schedules its own.  `build_par` lands here through the trait's default.
        Self::build(shapes)
    }

    fn traverse<'a, Query: IntersectsAabb<T, 3>, Shape: BHShape<T, 3>>(
        &'a self,
        query: &Query,
        shapes: &'a [Shape],
    ) -> Vec<&'a Shape> {
        // one generic query: the crate's own loop over the downloaded flat array (src/flat_bvh.rs:396-431);
        // ray BATCHES go through `traverse_batch`
        assert!(!self.flat_stale, "rebuild_async: call sync_flat() before a generic query");
        self.flat.traverse(query, shapes)
    }

    fn nearest_to<'a, Shape: BHShape<T, 3> + PointDistance<T, 3>>(
        &'a self,
        query: Point3<T>,
        shapes: &'a [Shape],
    ) -> Option<(&'a Shape, T)> {
        self.flat.nearest_to(query, shapes)
    }

    fn pretty_print(&self) {
        self.flat.pretty_print()
    }
}

impl<T: GpuScalar> Drop for GpuBvh<T> {
    fn drop(&mut self) {
        unsafe {
            ffi::bvhgpu_tree_destroy(self.tree);
            ffi::bvhgpu_destroy(self.ctx);
        }
    }
}

/// Traverse a crate-built `Bvh` on the GPU: `Bvh::flatten_custom` (src/flat_bvh.rs:240-251) lets the caller choose the node
/// type, so the `#[repr(C)]` layout of the C ABI needs no change to the crate.
pub fn upload_flat<T: GpuScalar>(bvh: &Bvh<T, 3>, shape_aabbs: &[[T; 6]], device: i32) -> (*mut ffi::bvhgpu_ctx, *mut ffi::bvhgpu_tree) {
    let flat: Vec<T::Flat> = bvh.flatten_custom(&|aabb: &Aabb<T, 3>, entry, exit, shape| T::flat_from_parts(aabb, entry, exit, shape));
    let mut ctx = core::ptr::null_mut();
    let mut tree = core::ptr::null_mut();
    unsafe {
        check(ctx, ffi::bvhgpu_create(device, core::ptr::null_mut(), &mut ctx));
        check(ctx, T::tree_from_flat(ctx, flat.as_ptr(), flat.len(), shape_aabbs.as_ptr().cast(), shape_aabbs.len(), &mut tree));
    }
    (ctx, tree)
}

/// `Ray::new` (src/ray/ray_impl.rs:70-80) — re-exported so that callers build rays the crate's way
pub fn ray<T: GpuScalar>(origin: [T; 3], direction: [T; 3]) -> Ray<T, 3> {
    Ray::new(Point3::from(origin), Vector3::from(direction))
}
