/*
 * bvh_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See bvh_oracle.h.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp; no -ffast-math)
 */
#include "bvh_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------- f32 instantiation ------------------------------- */
#define T float
#define S(x) x##_f32
#define NODE orc_node_f32
#define FLAT orc_flat_f32
#define RAY orc_ray_f32
#define T_EPS FLT_EPSILON
#define T_SQRT sqrtf
#include "oracle_impl.inc"
#undef T
#undef S
#undef NODE
#undef FLAT
#undef RAY
#undef T_EPS
#undef T_SQRT

/* ------------------------------- f64 instantiation ------------------------------- */
#define T double
#define S(x) x##_f64
#define NODE orc_node_f64
#define FLAT orc_flat_f64
#define RAY orc_ray_f64
#define T_EPS DBL_EPSILON
#define T_SQRT sqrt
#include "oracle_impl.inc"
#undef T
#undef S
#undef NODE
#undef FLAT
#undef RAY
#undef T_EPS
#undef T_SQRT

/* ------------------------------- testbase.rs generators ------------------------------- */

/* splitmix64 — testbase.rs:558-564 */
uint64_t orc_splitmix64(uint64_t *x) {
    *x += 0x9E3779B97F4A7C15ull;
    uint64_t z = *x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* next_point3_raw — testbase.rs:567-573 (a, b are i64; rotate_left is a 64-bit rotate) */
void orc_next_point3_raw(uint64_t *seed, int32_t out[3]) {
    uint64_t u = orc_splitmix64(seed);
    int64_t a = (int64_t)((u >> 32) & 0xFFFFFFFFull) - 0x80000000ll;
    int64_t b = (int64_t)(u & 0xFFFFFFFFull) - 0x80000000ll;
    uint64_t ub = (uint64_t)b;
    uint64_t rot = (ub << 6) | (ub >> 58);
    int64_t c = a ^ (int64_t)rot;
    out[0] = (int32_t)a;
    out[1] = (int32_t)b;
    out[2] = (int32_t)(uint32_t)(uint64_t)c; /* `as i32` truncates to the low 32 bits */
}

/* next_point3 — testbase.rs:576-595 */
void orc_next_point3(uint64_t *seed, const float bounds[6], float out[3]) {
    int32_t r[3];
    orc_next_point3_raw(seed, r);
    const float imax = (float)INT32_MAX; /* i32::MAX as f32 = 2147483648.0 */
    for (int k = 0; k < 3; k++) {
        float q = (float)r[k] / imax;
        float fv = (q + 1.0f) * 0.5f;
        float size = bounds[3 + k] - bounds[k];
        float off = fv * size;
        out[k] = bounds[k] + off;
    }
}

static void tri_aabb(const float *t, float *box) { /* Triangle::new — testbase.rs:325-333 */
    float b[6];
    aabb_empty_f32(b);
    aabb_grow_f32(b, t);
    aabb_grow_f32(b, t + 3);
    aabb_grow_f32(b, t + 6);
    memcpy(box, b, sizeof b);
}

/* push_cube — testbase.rs:490-554 (vertex order preserved) */
static void push_cube(const float pos[3], float *tris /* 12*9 */) {
    /* corner offsets: tfr, tbr, tbl, tfl, bfr, bbr, bbl, bfl */
    static const float off[8][3] = {
        { 0.5f, 0.5f, -0.5f }, { 0.5f, 0.5f, 0.5f }, { -0.5f, 0.5f, 0.5f }, { -0.5f, 0.5f, -0.5f },
        { 0.5f, -0.5f, -0.5f }, { 0.5f, -0.5f, 0.5f }, { -0.5f, -0.5f, 0.5f }, { -0.5f, -0.5f, -0.5f } };
    enum { TFR, TBR, TBL, TFL, BFR, BBR, BBL, BFL };
    static const int tri[12][3] = {
        { TBR, TFR, TFL }, { TFL, TBL, TBR }, { BFL, BFR, BBR }, { BBR, BBL, BFL },
        { TBL, TFL, BFL }, { BFL, BBL, TBL }, { BFR, TFR, TBR }, { TBR, BBR, BFR },
        { TFL, TFR, BFR }, { BFR, BFL, TFL }, { BBR, TBR, TBL }, { TBL, BBL, BBR } };
    float v[8][3];
    for (int c = 0; c < 8; c++)
        for (int k = 0; k < 3; k++) v[c][k] = pos[k] + off[c][k];
    for (int t = 0; t < 12; t++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) tris[t * 9 + j * 3 + k] = v[tri[t][j]][k];
}

/* create_n_cubes — testbase.rs:608-615 (seed 0) */
void orc_create_n_cubes(size_t n_cubes, const float bounds[6], float *tris, float *aabbs) {
    uint64_t seed = 0;
    for (size_t i = 0; i < n_cubes; i++) {
        float pos[3];
        orc_next_point3(&seed, bounds, pos);
        push_cube(pos, tris + i * 12 * 9);
        for (int t = 0; t < 12; t++) tri_aabb(tris + (i * 12 + t) * 9, aabbs + (i * 12 + t) * 6);
    }
}

/* create_ray — testbase.rs:687-691; the bench starts the stream at seed 0 (:825).
 * splitmix64's state after k draws is k*GAMMA, so ray `first` starts at state 2*first*GAMMA. */
void orc_create_rays(uint64_t first, size_t n, const float bounds[6], orc_ray_f32 *rays) {
    uint64_t seed = 2ull * first * 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; i++) {
        float o[3], d[3];
        orc_next_point3(&seed, bounds, o);
        orc_next_point3(&seed, bounds, d);
        orc_ray_new_f32(o, d, &rays[i]);
    }
}

/* Coherent primary rays for BASELINE.json configs[2].  The reference has no camera; this is the
 * engine's definition, restated here operation by operation (all f32, no FMA):
 *   sx = (((x + 0.5) / W) * 2) - 1,  sy = 1 - (((y + 0.5) / H) * 2)
 *   dir_k = (forward_k + (sx * tan_x) * right_k) + (sy * tan_y) * up_k,  ray = Ray::new(eye, dir)  (ray_impl.rs:70-80)
 * cam = eye[3], right[3], up[3], forward[3], tan_x, tan_y; ray index = y * W + x (row-major). */
void orc_primary_rays(const float cam[14], uint32_t width, uint32_t height, uint64_t first, size_t n, orc_ray_f32 *rays) {
    (void)height;
    for (size_t i = 0; i < n; i++) {
        const uint64_t id = first + i;
        const uint32_t x = (uint32_t)(id % width), y = (uint32_t)(id / width);
        float fx = (float)x + 0.5f; fx = fx / (float)width; fx = fx * 2.0f; const float sx = fx - 1.0f;
        float fy = (float)y + 0.5f; fy = fy / (float)height; fy = fy * 2.0f; const float sy = 1.0f - fy;
        const float ax = sx * cam[12], ay = sy * cam[13];
        float d[3];
        for (int k = 0; k < 3; k++) {
            const float r = ax * cam[3 + k], u = ay * cam[6 + k];
            float t = cam[9 + k] + r;
            d[k] = t + u;
        }
        orc_ray_new_f32(cam, d, &rays[i]);
    }
}

/* The f64 twins of the two streams, as the engine defines them for BASELINE.json configs[4]: the SAME f32 points
 * (origin and target / camera direction computed in f32 as above), widened to f64, then Ray::new in f64
 * (include/bvh_mi355x.h bvhgpu_gen_rays_f64 / bvhgpu_gen_primary_rays_f64). */
void orc_create_rays_f64(uint64_t first, size_t n, const float bounds[6], orc_ray_f64 *rays) {
    uint64_t seed = 2ull * first * 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; i++) {
        float o[3], d[3];
        orc_next_point3(&seed, bounds, o);
        orc_next_point3(&seed, bounds, d);
        const double oo[3] = { (double)o[0], (double)o[1], (double)o[2] }, dd[3] = { (double)d[0], (double)d[1], (double)d[2] };
        orc_ray_new_f64(oo, dd, &rays[i]);
    }
}
void orc_primary_rays_f64(const float cam[14], uint32_t width, uint32_t height, uint64_t first, size_t n, orc_ray_f64 *rays) {
    (void)height;
    for (size_t i = 0; i < n; i++) {
        const uint64_t id = first + i;
        const uint32_t x = (uint32_t)(id % width), y = (uint32_t)(id / width);
        float fx = (float)x + 0.5f; fx = fx / (float)width; fx = fx * 2.0f; const float sx = fx - 1.0f;
        float fy = (float)y + 0.5f; fy = fy / (float)height; fy = fy * 2.0f; const float sy = 1.0f - fy;
        const float ax = sx * cam[12], ay = sy * cam[13];
        double oo[3], dd[3];
        for (int k = 0; k < 3; k++) {
            const float r = ax * cam[3 + k], u = ay * cam[6 + k];
            float t = cam[9 + k] + r;
            const float d = t + u;
            oo[k] = (double)cam[k];
            dd[k] = (double)d;
        }
        orc_ray_new_f64(oo, dd, &rays[i]);
    }
}

/* intersect_bh — testbase.rs:819-837, the loop every "intersect" bench of the reference times, WHOLE:
 *     let ray = create_ray(&mut seed, bounds);          (:825; with `cam`: the engine's primary-ray definition above instead)
 *     let hits = bh.traverse(&ray, triangles);          (:828; FlatBvh::traverse, one walk into a growable Vec)
 *     for triangle in &hits { ray.intersects_triangle(&triangle.a, &triangle.b, &triangle.c); }     (:831-833)
 * for rays [first, first + n) of the stream, rays-parallel (splitmix64's state after k draws is k*GAMMA: every ray starts in O(1)).
 * bench.py's cpu_baseline of the harness entries times this.  Returns the number of candidates; *checksum (nullable) = sum of the
 * candidates' shape indices + the number of finite distances, so that nothing can be optimised away. */
uint64_t orc_harness_loop_f32(const orc_flat_f32 *flat, size_t n_flat, const float *shape_aabbs, const float *tris, uint64_t first,
                              size_t n_rays, const float bounds[6], const float *cam, uint32_t width, uint32_t height, int threads,
                              uint64_t *checksum) {
    uint64_t total = 0, sum = 0;
    if (threads < 1) threads = 1;
    (void)threads;
    const long long nr = (long long)n_rays;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads) reduction(+ : total, sum)
    for (long long i = 0; i < nr; i++) {
        orc_ray_f32 ray;
        if (cam) orc_primary_rays(cam, width, height, first + (uint64_t)i, 1, &ray);
        else orc_create_rays(first + (uint64_t)i, 1, bounds, &ray);
        uint32_t *vec = NULL;      /* Vec::new() */
        size_t len = 0, cap = 0;
        size_t index = 0;
        while (index < n_flat) {   /* flat_bvh.rs:408-428 */
            const orc_flat_f32 *nd = &flat[index];
            if (nd->entry == ORC_NONE) {
                if (ray_hit_f32(&ray, shape_aabbs + 6 * (size_t)nd->shape)) {
                    if (len == cap) { cap = cap ? 2 * cap : 4; vec = (uint32_t *)realloc(vec, cap * sizeof *vec); }
                    vec[len++] = nd->shape;
                }
                index = nd->exit;
            } else {
                float box[6] = { nd->min[0], nd->min[1], nd->min[2], nd->max[0], nd->max[1], nd->max[2] };
                index = ray_hit_f32(&ray, box) ? nd->entry : nd->exit;
            }
        }
        for (size_t k = 0; k < len; k++) {
            const float *t = tris + 9 * (size_t)vec[k];
            float uv[2];
            const float d = orc_ray_triangle_f32(&ray, t, t + 3, t + 6, uv);
            sum += vec[k] + (d < INFINITY ? 1u : 0u);
        }
        total += len;
        free(vec);
    }
    if (checksum) *checksum = sum;
    return total;
}

/* generate_aligned_boxes + UnitBox::aabb — testbase.rs:109-116, 84-89 */
void orc_aligned_boxes(float *aabbs) {
    int i = 0;
    for (int x = -10; x < 11; x++, i++) {
        float pos[3] = { (float)x, 0.0f, 0.0f };
        for (int k = 0; k < 3; k++) {
            aabbs[i * 6 + k] = pos[k] + -0.5f;
            aabbs[i * 6 + 3 + k] = pos[k] + 0.5f;
        }
    }
}
