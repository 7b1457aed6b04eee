//! cargo run --release -- <out_dir> [--full]
//! For create_n_cubes(100) and create_n_cubes(10_000) (the scenes of BASELINE.json configs[0] / configs[1]; generator restated
//! from src/testbase.rs:490-615 because `testbase` is a #[cfg(test)] module of the crate) this writes, little-endian:
//!   cubes{N}_aabbs.f32        n x [min xyz, max xyz]                      (Triangle::new's aabb, testbase.rs:325-333)
//!   cubes{N}_nodes.bin        2n-1 x bvhgpu_node_f32 (64 B)               Bvh::build
//!   cubes{N}_shape_nodes.u32  n                                           bh_node_index after the build
//!   cubes{N}_flat.bin         3n-2 x bvhgpu_flat_f32 (36 B)               Bvh::flatten_custom
//!   cubes{N}_rays.bin         R x bvhgpu_ray_f32 (36 B)                   create_ray(seed 0) stream, R = 1000 / 100000
//!   cubes{N}_offsets.u32, cubes{N}_indices.u32                            FlatBvh::traverse per ray, CSR
//!   manifest.json             sha256 of every array (the N = 10000 arrays themselves only with --full)
use bvh::aabb::{Aabb, Bounded};
use bvh::bounding_hierarchy::{BHShape, BoundingHierarchy};
use bvh::bvh::{Bvh, BvhNode};
use bvh::ray::Ray;
use nalgebra::{Point3, Vector3};
use sha2::{Digest, Sha256};
use std::io::Write;

// ---- src/testbase.rs:558-595 ----
fn splitmix64(x: &mut u64) -> u64 {
    *x = x.wrapping_add(0x9E3779B97F4A7C15);
    let mut z = *x;
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}
fn next_point3_raw(seed: &mut u64) -> (i32, i32, i32) {
    let u = splitmix64(seed);
    let a = ((u >> 32) & 0xFFFFFFFF) as i64 - 0x80000000;
    let b = (u & 0xFFFFFFFF) as i64 - 0x80000000;
    let c = a ^ b.rotate_left(6);
    (a as i32, b as i32, c as i32)
}
fn next_point3(seed: &mut u64, min: &Point3<f32>, max: &Point3<f32>) -> Point3<f32> {
    let (a, b, c) = next_point3_raw(seed);
    let float_vector = Vector3::new(
        (a as f32 / i32::MAX as f32 + 1.0) * 0.5,
        (b as f32 / i32::MAX as f32 + 1.0) * 0.5,
        (c as f32 / i32::MAX as f32 + 1.0) * 0.5,
    );
    let size = max - min;
    min + float_vector.component_mul(&size)
}

// ---- Triangle, src/testbase.rs:316-351 ----
struct Triangle {
    a: Point3<f32>,
    b: Point3<f32>,
    c: Point3<f32>,
    aabb: Aabb<f32, 3>,
    node_index: usize,
    id: u32,
}
impl Triangle {
    fn new(a: Point3<f32>, b: Point3<f32>, c: Point3<f32>, id: u32) -> Triangle {
        Triangle { a, b, c, aabb: Aabb::empty().grow(&a).grow(&b).grow(&c), node_index: 0, id }
    }
}
impl Bounded<f32, 3> for Triangle {
    fn aabb(&self) -> Aabb<f32, 3> {
        self.aabb
    }
}
impl BHShape<f32, 3> for Triangle {
    fn set_bh_node_index(&mut self, index: usize) {
        self.node_index = index;
    }
    fn bh_node_index(&self) -> usize {
        self.node_index
    }
}

// ---- push_cube, src/testbase.rs:490-554: 12 triangles, this vertex order ----
fn push_cube(pos: Point3<f32>, shapes: &mut Vec<Triangle>) {
    let top_front_right = pos + Vector3::new(0.5, 0.5, -0.5);
    let top_back_right = pos + Vector3::new(0.5, 0.5, 0.5);
    let top_back_left = pos + Vector3::new(-0.5, 0.5, 0.5);
    let top_front_left = pos + Vector3::new(-0.5, 0.5, -0.5);
    let bottom_front_right = pos + Vector3::new(0.5, -0.5, -0.5);
    let bottom_back_right = pos + Vector3::new(0.5, -0.5, 0.5);
    let bottom_back_left = pos + Vector3::new(-0.5, -0.5, 0.5);
    let bottom_front_left = pos + Vector3::new(-0.5, -0.5, -0.5);
    let mut t = |a, b, c| {
        let id = shapes.len() as u32;
        shapes.push(Triangle::new(a, b, c, id));
    };
    t(top_back_right, top_front_right, top_front_left);
    t(top_front_left, top_back_left, top_back_right);
    t(bottom_front_left, bottom_front_right, bottom_back_right);
    t(bottom_back_right, bottom_back_left, bottom_front_left);
    t(top_back_left, top_front_left, bottom_front_left);
    t(bottom_front_left, bottom_back_left, top_back_left);
    t(bottom_front_right, top_front_right, top_back_right);
    t(top_back_right, bottom_back_right, bottom_front_right);
    t(top_front_left, top_front_right, bottom_front_right);
    t(bottom_front_right, bottom_front_left, top_front_left);
    t(bottom_back_right, top_back_right, top_back_left);
    t(top_back_left, bottom_back_left, bottom_back_right);
}

fn f32s(v: &[f32]) -> Vec<u8> {
    v.iter().flat_map(|x| x.to_le_bytes()).collect()
}
fn u32s(v: &[u32]) -> Vec<u8> {
    v.iter().flat_map(|x| x.to_le_bytes()).collect()
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let out = args.get(1).cloned().unwrap_or_else(|| "tests/golden/crate".to_string());
    let full = args.iter().any(|a| a == "--full");
    std::fs::create_dir_all(&out).unwrap();
    let (bmin, bmax) = (Point3::new(-100_000.0f32, -100_000.0, -100_000.0), Point3::new(100_000.0f32, 100_000.0, 100_000.0)); // default_bounds, :598-603
    let mut manifest = String::from("{\n");
    for (n_cubes, n_rays) in [(100usize, 1000usize), (10_000, 100_000)] {
        // create_n_cubes, src/testbase.rs:608-615: seed 0
        let mut seed = 0u64;
        let mut shapes = Vec::new();
        for _ in 0..n_cubes {
            push_cube(next_point3(&mut seed, &bmin, &bmax), &mut shapes);
        }
        let aabbs: Vec<f32> = shapes.iter().flat_map(|s| [s.aabb.min.x, s.aabb.min.y, s.aabb.min.z, s.aabb.max.x, s.aabb.max.y, s.aabb.max.z]).collect();
        let bvh = Bvh::build(&mut shapes);
        // Vec<BvhNode> in the C-ABI layout (include/bvh_mi355x.h: bvhgpu_node_f32): leaf AABB fields zero, l = r = u32::MAX
        let mut nodes = Vec::<u8>::new();
        for nd in &bvh.nodes {
            match nd {
                BvhNode::Leaf { parent_index, shape_index } => {
                    nodes.extend(f32s(&[0.0; 12]));
                    nodes.extend(u32s(&[*parent_index as u32, u32::MAX, u32::MAX, *shape_index as u32]));
                }
                BvhNode::Node { parent_index, child_l_index, child_l_aabb, child_r_index, child_r_aabb } => {
                    let (l, r) = (child_l_aabb, child_r_aabb);
                    nodes.extend(f32s(&[l.min.x, l.min.y, l.min.z, l.max.x, l.max.y, l.max.z, r.min.x, r.min.y, r.min.z, r.max.x, r.max.y, r.max.z]));
                    nodes.extend(u32s(&[*parent_index as u32, *child_l_index as u32, *child_r_index as u32, u32::MAX]));
                }
            }
        }
        let shape_nodes: Vec<u32> = shapes.iter().map(|s| s.node_index as u32).collect();
        // FlatNode array through flatten_custom (src/flat_bvh.rs:240-251) in the C-ABI layout (bvhgpu_flat_f32, 36 B)
        let flat: Vec<Vec<u8>> = bvh.flatten_custom(&|aabb: &Aabb<f32, 3>, entry: u32, exit: u32, shape: u32| {
            let mut e = f32s(&[aabb.min.x, aabb.min.y, aabb.min.z, aabb.max.x, aabb.max.y, aabb.max.z]);
            e.extend(u32s(&[entry, exit, shape]));
            e
        });
        let flat: Vec<u8> = flat.into_iter().flatten().collect();
        // create_ray stream, src/testbase.rs:687-691, seed 0; FlatBvh::traverse per ray
        let flat_bvh = bvh.flatten();
        let mut rseed = 0u64;
        let (mut rays, mut offsets, mut indices) = (Vec::<u8>::new(), vec![0u32], Vec::<u32>::new());
        for _ in 0..n_rays {
            let origin = next_point3(&mut rseed, &bmin, &bmax);
            let direction = next_point3(&mut rseed, &bmin, &bmax).coords;
            let ray = Ray::new(origin, direction);
            rays.extend(f32s(&[ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z,
                               ray.inv_direction.x, ray.inv_direction.y, ray.inv_direction.z]));
            for s in flat_bvh.traverse(&ray, &shapes) {
                indices.push(s.id);
            }
            offsets.push(indices.len() as u32);
        }
        let small = n_cubes <= 100 || full;
        for (name, bytes, keep) in [
            ("aabbs.f32", f32s(&aabbs), small), ("nodes.bin", nodes, small), ("shape_nodes.u32", u32s(&shape_nodes), small),
            ("flat.bin", flat, small), ("rays.bin", rays, small), ("offsets.u32", u32s(&offsets), small), ("indices.u32", u32s(&indices), true),
        ] {
            let file = format!("cubes{n_cubes}_{name}");
            let digest = Sha256::digest(&bytes);
            manifest.push_str(&format!("  \"{file}\": {{\"bytes\": {}, \"sha256\": \"{:x}\"}},\n", bytes.len(), digest));
            if keep {
                std::fs::File::create(format!("{out}/{file}")).unwrap().write_all(&bytes).unwrap();
            }
        }
    }
    // the record sizes this program writes (field by field, above): tests/test_reference_bins.py compares them — and the field order,
    // read from this source — with sizeof / offsetof of include/bvh_mi355x.h, so a dump can never be checked against another layout
    manifest.push_str("  \"_meta\": {\"crate\": \"bvh 0.12.0\", \"layouts\": {\"bvhgpu_node_f32\": 64, \"bvhgpu_flat_f32\": 36, \"bvhgpu_ray_f32\": 36}}\n}\n");
    std::fs::write(format!("{out}/manifest.json"), manifest).unwrap();
    println!("wrote {out}/manifest.json");
}
