//! `GpuBvh`: the `bvh` crate's `BoundingHierarchy` (src/bounding_hierarchy.rs:89-336) on an MI355X, over the C ABI of
//! libbvh_mi355x.so (include/bvh_mi355x.h).  Build / flatten / batched traversal run on the GPU and give the arrays the
//! crate's own `Bvh::build` + `Bvh::flatten` + `FlatBvh::traverse` give, bit for bit; queries that need
//! user callbacks (`IntersectsAabb` for anything but rays, arbitrary `PointDistance`) run the crate's own loops over the
//! downloaded flat array.
//!
//! NOTE: written against bvh 0.12.0 / nalgebra 0.34 by reading their sources; the image this engine was developed in has
//! no cargo/rustc, so this crate has NOT been compiled there.
#![allow(clippy::missing_safety_doc)]
pub mod ffi;

use bvh::aabb::{Aabb, IntersectsAabb};
use bvh::bounding_hierarchy::{BHShape, BoundingHierarchy};
use bvh::bvh::{Bvh, BvhNode, BvhNodeBuildArgs};
use bvh::flat_bvh::{FlatBvh, FlatNode};
use bvh::point_query::PointDistance;
use bvh::ray::Ray;
use core::ffi::c_int;
use nalgebra::{Point3, Vector3};

/// Panics with the engine's message: the crate has no error type, contract violations panic there too
/// (e.g. NaN centroids, src/bvh/bvh_node.rs:214-217).
fn check(ctx: *const ffi::bvhgpu_ctx, rc: c_int) {
    if rc != ffi::BVHGPU_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(ffi::bvhgpu_last_error(ctx)) };
        panic!("bvh_mi355x: status {rc}: {}", msg.to_string_lossy());
    }
}

fn aabb_to_6(b: &Aabb<f32, 3>) -> [f32; 6] {
    [b.min.x, b.min.y, b.min.z, b.max.x, b.max.y, b.max.z]
}

pub fn ray_to_ffi(r: &Ray<f32, 3>) -> ffi::bvhgpu_ray_f32 {
    // Ray { origin, direction, inv_direction } (src/ray/ray_impl.rs:17-29): copied field by field, never transmuted
    ffi::bvhgpu_ray_f32 {
        o: [r.origin.x, r.origin.y, r.origin.z],
        d: [r.direction.x, r.direction.y, r.direction.z],
        inv: [r.inv_direction.x, r.inv_direction.y, r.inv_direction.z],
    }
}

pub struct GpuBvh {
    ctx: *mut ffi::bvhgpu_ctx,
    tree: *mut ffi::bvhgpu_tree,
    n_shapes: usize,
    /// CPU copy of the flat array in the crate's own layout, for the generic queries of the trait
    flat: FlatBvh<f32, 3>,
}

// the handles are only used through &self / &mut self; the engine's ctx is not internally locked: one GpuBvh per thread
unsafe impl Send for GpuBvh {}

/// CSR result of a batch: ray i hit `indices[offsets[i]..offsets[i+1]]`, in the order `FlatBvh::traverse` returns them
pub struct BatchHits {
    pub offsets: Vec<u32>,
    pub indices: Vec<u32>,
}

impl GpuBvh {
    /// Bvh::build_par + Bvh::flatten on the GPU from the shapes' AABBs (n x [min xyz, max xyz])
    pub fn from_aabbs(aabbs: &[[f32; 6]], device: i32) -> GpuBvh {
        let mut ctx = core::ptr::null_mut();
        let mut tree = core::ptr::null_mut();
        unsafe {
            check(ctx, ffi::bvhgpu_create(device, core::ptr::null_mut(), &mut ctx));
            check(ctx, ffi::bvhgpu_build_flat_f32(ctx, aabbs.as_ptr().cast(), aabbs.len(), ffi::BVHGPU_HOST, &mut tree));
        }
        let mut me = GpuBvh { ctx, tree, n_shapes: aabbs.len(), flat: Vec::new() };
        me.flat = me.download_flat();
        me
    }

    /// the argument of `BHShape::set_bh_node_index` for every shape (src/bvh/bvh_node.rs:102)
    pub fn shape_nodes(&self) -> Vec<u32> {
        let mut sn = vec![0u32; self.n_shapes];
        unsafe { check(self.ctx, ffi::bvhgpu_tree_shape_nodes(self.tree, sn.as_mut_ptr(), ffi::BVHGPU_HOST)); }
        sn
    }

    /// `Vec<BvhNode>` exactly as `Bvh::build` produces it (bit-identical AABBs, same indices)
    pub fn to_bvh(&self) -> Bvh<f32, 3> {
        let nn = if self.n_shapes == 0 { 0 } else { 2 * self.n_shapes - 1 };
        let mut raw = vec![ffi::bvhgpu_node_f32::default(); nn];
        unsafe { check(self.ctx, ffi::bvhgpu_tree_nodes(self.tree, raw.as_mut_ptr().cast(), ffi::BVHGPU_HOST)); }
        let nodes = raw
            .iter()
            .map(|r| {
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
            })
            .collect();
        Bvh { nodes }
    }

    fn download_flat(&self) -> FlatBvh<f32, 3> {
        let nf = if self.n_shapes >= 2 { 3 * self.n_shapes - 2 } else { self.n_shapes };
        let mut raw = vec![ffi::bvhgpu_flat_f32::default(); nf];
        unsafe { check(self.ctx, ffi::bvhgpu_flat_nodes(self.tree, raw.as_mut_ptr().cast(), ffi::BVHGPU_HOST)); }
        raw.iter()
            .map(|f| FlatNode {
                aabb: Aabb::with_bounds(Point3::from(f.min), Point3::from(f.max)),
                entry_index: f.entry,
                exit_index: f.exit,
                shape_index: f.shape,
            })
            .collect()
    }

    /// `FlatBvh::traverse` (src/flat_bvh.rs:396-431) for many rays at once — what the GPU is for
    pub fn traverse_batch(&self, rays: &[Ray<f32, 3>]) -> BatchHits {
        let r: Vec<ffi::bvhgpu_ray_f32> = rays.iter().map(ray_to_ffi).collect();
        let mut hits = core::ptr::null_mut();
        let mut total = 0u64;
        unsafe {
            check(self.ctx, ffi::bvhgpu_traverse_f32(self.tree, r.as_ptr(), r.len(), ffi::BVHGPU_HOST, 0, &mut hits));
            check(self.ctx, ffi::bvhgpu_hits_info(hits, core::ptr::null_mut(), &mut total, core::ptr::null_mut()));
        }
        let (mut offsets, mut indices) = (vec![0u32; rays.len() + 1], vec![0u32; total as usize]);
        unsafe {
            check(self.ctx, ffi::bvhgpu_hits_fetch(hits, offsets.as_mut_ptr(), indices.as_mut_ptr(), core::ptr::null_mut(), ffi::BVHGPU_HOST));
            ffi::bvhgpu_hits_destroy(hits);
        }
        BatchHits { offsets, indices }
    }

    /// the shapes moved, the topology stays: `Bvh::fix_aabbs_ascending` (src/bvh/optimization.rs:355-391) over the whole tree
    pub fn refit(&mut self, aabbs: &[[f32; 6]]) {
        unsafe { check(self.ctx, ffi::bvhgpu_refit_f32(self.tree, aabbs.as_ptr().cast(), aabbs.len(), ffi::BVHGPU_HOST)); }
        self.flat = self.download_flat();
    }

    pub fn raw(&self) -> (*mut ffi::bvhgpu_ctx, *mut ffi::bvhgpu_tree) {
        (self.ctx, self.tree)
    }
}

impl BoundingHierarchy<f32, 3> for GpuBvh {
    fn build<Shape: BHShape<f32, 3>>(shapes: &mut [Shape]) -> GpuBvh {
        // (1) the only callback the device cannot make: shape.aabb(), gathered once per shape
        let aabbs: Vec<[f32; 6]> = shapes.iter().map(|s| aabb_to_6(&s.aabb())).collect();
        // (2) Bvh::build_par + Bvh::flatten on the GPU
        let bh = GpuBvh::from_aabbs(&aabbs, 0);
        // (3) set_bh_node_index for every shape (src/bvh/bvh_node.rs:102)
        for (s, ni) in shapes.iter_mut().zip(bh.shape_nodes()) {
            s.set_bh_node_index(ni as usize);
        }
        bh
    }

    fn build_with_executor<
        Shape: BHShape<f32, 3>,
        Executor: FnMut(BvhNodeBuildArgs<'_, Shape, f32, 3>, BvhNodeBuildArgs<'_, Shape, f32, 3>),
    >(
        shapes: &mut [Shape],
        _executor: Executor,
    ) -> GpuBvh {
        // the executor only schedules CPU sub-builds and cannot change the result (node placement is arithmetic,
        // src/bvh/bvh_node.rs:138-142): the GPU schedules its own.  `build_par` lands here through the trait's default.
        Self::build(shapes)
    }

    fn traverse<'a, Query: IntersectsAabb<f32, 3>, Shape: BHShape<f32, 3>>(
        &'a self,
        query: &Query,
        shapes: &'a [Shape],
    ) -> Vec<&'a Shape> {
        // one generic query: the crate's own loop over the downloaded flat array (src/flat_bvh.rs:396-431);
        // ray BATCHES go through `traverse_batch`
        self.flat.traverse(query, shapes)
    }

    fn nearest_to<'a, Shape: BHShape<f32, 3> + PointDistance<f32, 3>>(
        &'a self,
        query: Point3<f32>,
        shapes: &'a [Shape],
    ) -> Option<(&'a Shape, f32)> {
        self.flat.nearest_to(query, shapes)
    }

    fn pretty_print(&self) {
        self.flat.pretty_print()
    }
}

impl Drop for GpuBvh {
    fn drop(&mut self) {
        unsafe {
            ffi::bvhgpu_tree_destroy(self.tree);
            ffi::bvhgpu_destroy(self.ctx);
        }
    }
}

/// Traverse a crate-built `Bvh` on the GPU: `Bvh::flatten_custom` (src/flat_bvh.rs:240-251) lets the caller choose the node
/// type, so the `#[repr(C)]` layout of the C ABI needs no change to the crate.
pub fn upload_flat(bvh: &Bvh<f32, 3>, shape_aabbs: &[[f32; 6]], device: i32) -> (*mut ffi::bvhgpu_ctx, *mut ffi::bvhgpu_tree) {
    let flat: Vec<ffi::bvhgpu_flat_f32> = bvh.flatten_custom(&|aabb: &Aabb<f32, 3>, entry, exit, shape| ffi::bvhgpu_flat_f32 {
        min: [aabb.min.x, aabb.min.y, aabb.min.z],
        max: [aabb.max.x, aabb.max.y, aabb.max.z],
        entry,
        exit,
        shape,
    });
    let mut ctx = core::ptr::null_mut();
    let mut tree = core::ptr::null_mut();
    unsafe {
        check(ctx, ffi::bvhgpu_create(device, core::ptr::null_mut(), &mut ctx));
        check(ctx, ffi::bvhgpu_tree_from_flat_f32(ctx, flat.as_ptr(), flat.len(), shape_aabbs.as_ptr().cast(), shape_aabbs.len(), &mut tree));
    }
    (ctx, tree)
}

/// `Ray::new` (src/ray/ray_impl.rs:70-80) — re-exported so that callers build rays the crate's way
pub fn ray(origin: [f32; 3], direction: [f32; 3]) -> Ray<f32, 3> {
    Ray::new(Point3::from(origin), Vector3::from(direction))
}
