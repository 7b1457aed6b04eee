// common.hpp — shared device/host definitions of the MI355X BVH engine (gfx950, wave64).
//
// Arithmetic discipline (DESIGN.md §"Floating point"): every + - * / sqrt on the path is ONE
// correctly rounded IEEE-754 operation, exactly like the Rust reference.  This file is compiled
// with -ffp-contract=off (no FMA contraction) and without fast-math; hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt keeps `/` and sqrt correctly rounded.  min/max
// reductions are exact, so they may be re-associated freely (atomics, scans, shuffles).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cfloat>
#include <cmath>

#include "../../include/bvh_mi355x.h"

namespace bvhgpu {

constexpr uint32_t NONE = BVHGPU_NONE;
constexpr int NUM_BUCKETS = 6;        // reference: src/bvh/bucket.rs:5
constexpr int WAVE = 64;              // gfx950 wavefront
constexpr int SMALL_MAX = 64;         // segments <= one wave are finished by the wave-subtree kernel
#ifndef BVH_TILE
#define BVH_TILE 512
#endif
constexpr int TILE = BVH_TILE;        // positions per workgroup tile in the level-synchronous tier (256 / 512 / 1024 measured: 512)
constexpr int STAT_KEYS = 12;         // aabb min3,max3, centroid min3,max3

// ------------------------------------------------------------------------------------------------
// per-scalar-type traits
// ------------------------------------------------------------------------------------------------
template <typename T> struct Traits;

template <> struct Traits<float> {
    using Key = uint32_t;
    using Node = bvhgpu_node_f32;
    using Flat = bvhgpu_flat_f32;
    using Ray = bvhgpu_ray_f32;
    static constexpr int dtype = BVHGPU_F32;
    __host__ __device__ static constexpr float eps() { return FLT_EPSILON; }  // T::epsilon(), bvh_node.rs:114
    __host__ __device__ static float inf() { return INFINITY; }
    // monotone float -> unsigned map (IEEE-754-2019 total order on NaN-free data: -0 < +0)
    __host__ __device__ static Key key(float f) {
        uint32_t b;
        __builtin_memcpy(&b, &f, 4);
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    __host__ __device__ static float unkey(Key k) {
        uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
        float f;
        __builtin_memcpy(&f, &b, 4);
        return f;
    }
    static constexpr Key KEY_POS_INF = 0xFF800000u;  // key(+inf): identity of min
    static constexpr Key KEY_NEG_INF = 0x007FFFFFu;  // key(-inf): identity of max
};

template <> struct Traits<double> {
    using Key = unsigned long long;
    using Node = bvhgpu_node_f64;
    using Flat = bvhgpu_flat_f64;
    using Ray = bvhgpu_ray_f64;
    static constexpr int dtype = BVHGPU_F64;
    __host__ __device__ static constexpr double eps() { return DBL_EPSILON; }
    __host__ __device__ static double inf() { return INFINITY; }
    __host__ __device__ static Key key(double f) {
        unsigned long long b;
        __builtin_memcpy(&b, &f, 8);
        return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
    }
    __host__ __device__ static double unkey(Key k) {
        unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
        double f;
        __builtin_memcpy(&f, &b, 8);
        return f;
    }
    static constexpr Key KEY_POS_INF = 0xFFF0000000000000ull;
    static constexpr Key KEY_NEG_INF = 0x000FFFFFFFFFFFFFull;
};

// ------------------------------------------------------------------------------------------------
// engine-private traversal node ("folded" pre-order array, DESIGN.md §"Traversal layout"):
// one entry per non-root tree node; on a slab hit go to i+1, on a miss go to `exit`; an entry
// whose `shape` != NONE reports that shape when hit.  32 B (f32) / 64 B (f64), naturally aligned
// so a lane fetches it with two (four) 16-byte loads.
// ------------------------------------------------------------------------------------------------
template <typename T> struct TravNode;
template <> struct __attribute__((aligned(32))) TravNode<float> {
    float mn[3];
    uint32_t exit;
    float mx[3];
    uint32_t shape;
};
template <> struct __attribute__((aligned(64))) TravNode<double> {
    double mn[3];
    double mx[3];
    uint32_t exit;
    uint32_t shape;
    uint64_t _pad;
};
// `shape` word of a traversal entry: a leaf holds its shape index (< 2^31: MAX_SHAPES is (2^32-2)/3);
// an inner entry has TRAV_INNER set and carries in its low 16 bits the LDS slot of its EXIT entry
// (SLOT_NONE if that entry is not one of the top-of-tree entries traversal keeps in LDS).
constexpr uint32_t TRAV_INNER = 0x80000000u;
constexpr uint32_t SLOT_NONE = 0xFFFFu;
__host__ __device__ inline bool trav_is_leaf(uint32_t w) { return (w & TRAV_INNER) == 0u; }
__host__ __device__ inline uint32_t heap_child(uint32_t h, uint32_t side) {  // saturating 2h + side
    const uint32_t c = 2u * h + side;
    return (h >= 0x8000u || c > SLOT_NONE) ? SLOT_NONE : c;
}
// top-of-tree slots = heap numbers (root 1, children 2h / 2h+1) below TopCfg<T>::SLOTS; what fits 160 KB of LDS
template <typename T> struct TopCfg;
#ifndef BVH_TOP_SLOTS_F32
#define BVH_TOP_SLOTS_F32 5056
#endif
template <> struct TopCfg<float> { static constexpr uint32_t SLOTS = BVH_TOP_SLOTS_F32; };   // 2 x 16 B per slot
template <> struct TopCfg<double> { static constexpr uint32_t SLOTS = 2880; };  // 56 B per slot
static_assert(sizeof(TravNode<float>) == 32, "trav f32");
static_assert(sizeof(TravNode<double>) == 64, "trav f64");

// ------------------------------------------------------------------------------------------------
// engine-private WIDE node (DESIGN.md §"Wide walk"): for every inner tree node b, the boxes of its (up to) four
// GRANDCHILDREN in left-to-right order — slots 0,1 = the children of b's left child, slots 2,3 = those of its right
// child; a child that is itself a leaf occupies the first slot of its pair (its box IS the shape's AABB) and leaves the
// second one absent.  SoA over the four slots, one 16-byte chunk per coordinate, so a lane fetches a node with 7 (f32) /
// 13 (f64) 16-byte loads from one 128- / 256-byte line.  ref: a leaf holds its shape index (< 2^31); an inner grandchild
// holds WIDE_INNER | its tree node index (it has a wide node of its own); an absent slot holds NONE and a NaN box (which
// fails every slab test).  Why skipping b's children is exact: see traverse.hip.
// ------------------------------------------------------------------------------------------------
template <typename T> struct WideNode;
template <> struct __attribute__((aligned(128))) WideNode<float> {
    float mn[3][4], mx[3][4];
    uint32_t ref[4];
    uint32_t _pad[4];
};
template <> struct __attribute__((aligned(256))) WideNode<double> {
    double mn[3][4], mx[3][4];
    uint32_t ref[4];
    uint32_t _pad[12];
};
static_assert(sizeof(WideNode<float>) == 128, "wide f32");
static_assert(sizeof(WideNode<double>) == 256, "wide f64");
// ---- guide boxes of f64 trees (engine.hpp bvhgpu_tree::wide_guide) -------------------------------------------------------------
// A guide box is the f64 box grown by delta = GUIDE_GROW * S on every side (S = largest |coordinate| of the scene) and rounded outward
// to f32.  For a ray with |o_k| <= GUIDE_ORIGIN_MAX * S and |inv_k| * S inside the f32 range, the f32 slab test on the guide box with
// the round-to-nearest f32 ray passes whenever the f64 test on the f64 box does: every f32 rounding (origin, inverse direction,
// difference, product) moves a plane's t by less than |inv_k| * 2^-21 * (S + |o_k|), the growth moves it by |inv_k| * 2^-18 * S the other way
// (tests/test_guide_cpu.py replays the argument numerically on grazing rays).  Rays outside the range are never walked this way.
// Worst case, every rounding adversarial (o32 by 2^-24 * 3S, the difference, 1/d and the product by 2^-24 each of a quantity <= 4S(1 + 2^-18)):
// the near plane's t moves the wrong way by less than |inv_k| * 15 * 2^-24 * S against a growth of |inv_k| * 64 * 2^-24 * S — a factor 4.2;
// tests/test_guide_cpu.py checks it with DIRECTED roundings (1 ulp each instead of 1/2: 30 against 64) at the edges of the range.
constexpr double GUIDE_GROW = 1.0 / 262144.0;   // 2^-18
constexpr double GUIDE_ORIGIN_MAX = 3.0;
constexpr double GUIDE_SCENE_MIN = 0x1p-125;               // smallest S the argument covers (f32 denormal spacing is 2^-149)
constexpr double GUIDE_SCENE_MAX = 0x1p125;                // largest: a difference b32 - o32 reaches 4 S (1 + 2^-18) and must stay finite in f32
                                                           // (found by the directed-rounding test: at S = 0.999 FLT_MAX it overflowed to inf)
constexpr double GUIDE_F32_MAX = 0x1.fffffep127;           // FLT_MAX
constexpr double GUIDE_F32_MIN_NORMAL = 0x1p-126;          // FLT_MIN
__host__ __device__ inline float f32_below(double x) {   // largest float <= x (NaN stays NaN)
    float f = (float)x;
    if ((double)f > x) {
        uint32_t b; __builtin_memcpy(&b, &f, 4);
        b = f > 0.0f ? b - 1u : (f < 0.0f ? b + 1u : 0x80000001u);
        __builtin_memcpy(&f, &b, 4);
    }
    return f;
}
__host__ __device__ inline float f32_above(double x) { return -f32_below(-x); }
constexpr uint32_t WIDE_INNER = 0x80000000u;
constexpr uint32_t WIDE_RESIDENT = 0x40000000u;        // walk-private: WIDE_INNER | WIDE_RESIDENT | LDS slot (4-ary heap number)
constexpr size_t WIDE_MAX_SHAPES = (size_t)1 << 28;    // node indices fit 30 bits, per-ray hit counts fit 28 bits
constexpr uint32_t WIDE_SLOTS = 1365;                  // 4-ary heap slots of wide levels 0..5 (tree levels 0,2,..,10): the most LDS can hold
// wide-level k of 4-ary heap slot q (root 0, children 4q+1..4q+4) starts at slot (4^k - 1) / 3
__host__ __device__ inline uint32_t wide_level_base(int k) { return ((1u << (2 * k)) - 1u) / 3u; }

// work item of the builder: one tree node still to be split (BvhNodeBuildArgs, bvh_node.rs:437-446)
template <typename T> struct Item {
    uint32_t ni;       // node_index
    uint32_t parent;   // parent_index
    uint32_t start;    // first position of its index slice
    uint32_t count;    // indices.len()
    uint32_t tile_base;
    uint32_t parity;   // which idx buffer holds its slice
    uint32_t heap;     // heap number of the node (root 1, children 2h / 2h+1), saturated at SLOT_NONE
    uint32_t _r1;
    T A[6];            // aabb_bounds
    T C[6];            // centroid_bounds
};

// ------------------------------------------------------------------------------------------------
// exact Aabb helpers (each cites src/aabb/aabb_impl.rs)
// ------------------------------------------------------------------------------------------------
template <typename T> __host__ __device__ inline T tmin(T a, T b) {  // inf (:305); -0 < +0
    return (a < b || (a == b && signbit(a))) ? a : b;
}
template <typename T> __host__ __device__ inline T tmax(T a, T b) {  // sup (:306)
    return (a > b || (a == b && !signbit(a))) ? a : b;
}
// center = min*0.5 + max*0.5 (:501-504) — two multiplies and one add, never (min+max)*0.5
template <typename T> __host__ __device__ inline T center1(T mn, T mx) {
    T lo = mn * (T)0.5;
    T hi = mx * (T)0.5;
    return lo + hi;
}
// surface_area = 2 * ((sx*sx + sy*sy) + sz*sz) (:551-554 with nalgebra's 3-vector dot)
template <typename T> __host__ __device__ inline T surface_area(const T* b) {
    T sx = b[3] - b[0], sy = b[4] - b[1], sz = b[5] - b[2];
    T xx = sx * sx, yy = sy * sy, zz = sz * sz;
    T d = xx + yy;
    d = d + zz;
    return (T)2 * d;
}
// largest_axis = size.imax(): first strict maximum (:594-596)
template <typename T> __host__ __device__ inline int largest_axis(const T* b) {
    T s0 = b[3] - b[0], s1 = b[4] - b[1], s2 = b[5] - b[2];
    int a = 0;
    T m = s0;
    if (s1 > m) { m = s1; a = 1; }
    if (s2 > m) { m = s2; a = 2; }
    return a;
}
// bucket index (bvh_node.rs:210-217): trunc(((c - cmin) / ext) * (T(6) - T(0.01)))
template <typename T> __host__ __device__ inline int bucket_of(T c, T cmin, T ext) {
    const T K = (T)NUM_BUCKETS - (T)0.01;
    T rel = (c - cmin) / ext;
    T scaled = rel * K;
    return (int)scaled;
}

// ------------------------------------------------------------------------------------------------
// ray / AABB slab test (src/ray/intersect_default.rs:16-37) — bit-exact boolean
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ bool slab_hit(const T o[3], const T inv[3], const T mn[3], const T mx[3], T& tmin_out,
                                         T& tmax_out) {
    T l0 = (mn[0] - o[0]) * inv[0], h0 = (mx[0] - o[0]) * inv[0];
    T l1 = (mn[1] - o[1]) * inv[1], h1 = (mx[1] - o[1]) * inv[1];
    T l2 = (mn[2] - o[2]) * inv[2], h2 = (mx[2] - o[2]) * inv[2];
    // has_nan(lbr) | has_nan(rtr) -> miss (:22-28).  x != x is the NaN test.
    bool nan = (l0 != l0) | (h0 != h0) | (l1 != l1) | (h1 != h1) | (l2 != l2) | (h2 != h2);
    // NaN-free from here: min/max are exact and order-free
    T a0 = l0 < h0 ? l0 : h0, b0 = l0 < h0 ? h0 : l0;
    T a1 = l1 < h1 ? l1 : h1, b1 = l1 < h1 ? h1 : l1;
    T a2 = l2 < h2 ? l2 : h2, b2 = l2 < h2 ? h2 : l2;
    T tmn = a0 > a1 ? a0 : a1;
    tmn = tmn > a2 ? tmn : a2;
    T tmx = b0 < b1 ? b0 : b1;
    tmx = tmx < b2 ? tmx : b2;
    T z = tmn > (T)0 ? tmn : (T)0;  // fast_max(tmin, 0) (utils.rs:52-54)
    tmin_out = z;                   // intersection_slice_for_aabb's tmin (ray_impl.rs:135)
    tmax_out = tmx;
    return !nan && (tmx >= z);
}

// The same boolean for a ray whose origin and inv_direction are all finite and a NaN-free box: then no
// l/h can be NaN (inf - finite = inf, inf * finite-nonzero = inf; |inv| >= 1 for a normalised direction),
// the NaN branch (:22-28) is dead, and on NaN-free values IEEE minNum/maxNum return the same VALUE as
// the reference's compare-selects (only the sign of a zero result can differ, which no comparison sees).
// v_min_f32 / v_max_f32 / v_max3_f32 / v_min3_f32: 22 VALU instead of 43.  Not used when the t-slice is
// returned (sign of zero), nor for rays with a non-finite component (wave-uniform fallback to slab_hit).
template <typename T>
__device__ __forceinline__ bool slab_hit_finite(const T o[3], const T inv[3], const T mn[3], const T mx[3]) {
    T l0 = (mn[0] - o[0]) * inv[0], h0 = (mx[0] - o[0]) * inv[0];
    T l1 = (mn[1] - o[1]) * inv[1], h1 = (mx[1] - o[1]) * inv[1];
    T l2 = (mn[2] - o[2]) * inv[2], h2 = (mx[2] - o[2]) * inv[2];
    T tmn = fmax(fmax(fmin(l0, h0), fmin(l1, h1)), fmin(l2, h2));
    T tmx = fmin(fmin(fmax(l0, h0), fmax(l1, h1)), fmax(l2, h2));
    return tmx >= fmax(tmn, (T)0);
}
// f32: the instructions are written out — through fminf/fmaxf the compiler adds one canonicalising
// v_max_f32 x,x per product (sNaN quieting that NaN-free data does not need)
template <>
__device__ __forceinline__ bool slab_hit_finite<float>(const float o[3], const float inv[3], const float mn[3],
                                                       const float mx[3]) {
    float l0 = (mn[0] - o[0]) * inv[0], h0 = (mx[0] - o[0]) * inv[0];
    float l1 = (mn[1] - o[1]) * inv[1], h1 = (mx[1] - o[1]) * inv[1];
    float l2 = (mn[2] - o[2]) * inv[2], h2 = (mx[2] - o[2]) * inv[2];
    float a0, a1, a2, b0, b1, b2, tmn, tmx;
    asm("v_min_f32 %0, %1, %2" : "=v"(a0) : "v"(l0), "v"(h0));
    asm("v_max_f32 %0, %1, %2" : "=v"(b0) : "v"(l0), "v"(h0));
    asm("v_min_f32 %0, %1, %2" : "=v"(a1) : "v"(l1), "v"(h1));
    asm("v_max_f32 %0, %1, %2" : "=v"(b1) : "v"(l1), "v"(h1));
    asm("v_min_f32 %0, %1, %2" : "=v"(a2) : "v"(l2), "v"(h2));
    asm("v_max_f32 %0, %1, %2" : "=v"(b2) : "v"(l2), "v"(h2));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmn) : "v"(a0), "v"(a1), "v"(a2));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmx) : "v"(b0), "v"(b1), "v"(b2));
    return (tmx >= tmn) & (tmx >= 0.0f);   // tmx >= max(tmn, 0)
}
// the same test, also returning how long the ray stays inside the box (scheduling heuristics only)
template <typename T>
__device__ __forceinline__ bool slab_hit_finite_len(const T o[3], const T inv[3], const T mn[3], const T mx[3], T& len) {
    T l0 = (mn[0] - o[0]) * inv[0], h0 = (mx[0] - o[0]) * inv[0];
    T l1 = (mn[1] - o[1]) * inv[1], h1 = (mx[1] - o[1]) * inv[1];
    T l2 = (mn[2] - o[2]) * inv[2], h2 = (mx[2] - o[2]) * inv[2];
    T tmn = fmax(fmax(fmin(l0, h0), fmin(l1, h1)), fmin(l2, h2));
    T tmx = fmin(fmin(fmax(l0, h0), fmax(l1, h1)), fmax(l2, h2));
    const T t0 = fmax(tmn, (T)0);
    len = tmx - t0;
    return tmx >= t0;
}
template <>
__device__ __forceinline__ bool slab_hit_finite_len<float>(const float o[3], const float inv[3], const float mn[3], const float mx[3],
                                                           float& len) {
    float l0 = (mn[0] - o[0]) * inv[0], h0 = (mx[0] - o[0]) * inv[0];
    float l1 = (mn[1] - o[1]) * inv[1], h1 = (mx[1] - o[1]) * inv[1];
    float l2 = (mn[2] - o[2]) * inv[2], h2 = (mx[2] - o[2]) * inv[2];
    float a0, a1, a2, b0, b1, b2, tmn, tmx, t0;
    asm("v_min_f32 %0, %1, %2" : "=v"(a0) : "v"(l0), "v"(h0));
    asm("v_max_f32 %0, %1, %2" : "=v"(b0) : "v"(l0), "v"(h0));
    asm("v_min_f32 %0, %1, %2" : "=v"(a1) : "v"(l1), "v"(h1));
    asm("v_max_f32 %0, %1, %2" : "=v"(b1) : "v"(l1), "v"(h1));
    asm("v_min_f32 %0, %1, %2" : "=v"(a2) : "v"(l2), "v"(h2));
    asm("v_max_f32 %0, %1, %2" : "=v"(b2) : "v"(l2), "v"(h2));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmn) : "v"(a0), "v"(a1), "v"(a2));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmx) : "v"(b0), "v"(b1), "v"(b2));
    asm("v_max_f32 %0, 0, %1" : "=v"(t0) : "v"(tmn));
    len = tmx - t0;
    return tmx >= t0;
}
template <typename T> __device__ __forceinline__ bool ray_is_finite(const T o[3], const T inv[3]) {
    bool f = true;
#pragma unroll
    for (int k = 0; k < 3; k++) f = f && (fabs(o[k]) < Traits<T>::inf()) && (fabs(inv[k]) < Traits<T>::inf());
    return f;   // false for inf and for NaN
}

// ------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------
// Aabb join on NaN-free floats in ONE instruction: V_MIN_F32 / V_MAX_F32 (and the F64 forms) order
// -0 < +0 (ISA: "if S0 == +0 and S1 == -0 return S1"), i.e. exactly tmin / tmax above and the integer-key
// order.  Written as asm so that no canonicalising v_max x,x is added for values that come out of shuffles.
__device__ __forceinline__ float join_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float join_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double join_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double join_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// value of lane (byte address `addr4` = 4 * lane) — one ds_bpermute_b32 per dword
__device__ __forceinline__ float lane_fetch(float v, int addr4) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(addr4, __float_as_int(v)));
}
__device__ __forceinline__ double lane_fetch(double v, int addr4) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr4, (int)(b & 0xFFFFFFFFll));
    const int hi = __builtin_amdgcn_ds_bpermute(addr4, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// DPP lane moves: the value of the lane the control selects, or the lane's OWN value where that lane is outside the row /
// masked (so that join(x, fetched) is a no-op there).  No LDS crossbar trip, unlike ds_bpermute.
// controls: row_shr:n = 0x110 + n, row_shl:n = 0x100 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_fetch(float v) {
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(b, b, CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ double dpp_fetch(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = (int)(b & 0xFFFFFFFFll), hi = (int)(b >> 32);
    const int l2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    const int h2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double(((long long)h2 << 32) | (unsigned int)l2);
}
// the same without a defined value for lanes whose source lies outside the row (or whose row the mask leaves out): the caller masks those lanes itself (no copy of v
// into the destination first — the segmented scans of the builder's wave tier are bound by issue slots)
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_fetch_raw(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ double dpp_fetch_raw(double v) {
    const long long b = __double_as_longlong(v);
    const int l2 = __builtin_amdgcn_mov_dpp((int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xF, false);
    const int h2 = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double(((long long)h2 << 32) | (unsigned int)l2);
}
// value of one (compile-time) lane in every lane, through an SGPR
template <int LANE> __device__ __forceinline__ float lane_bcast(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), LANE)); }
template <int LANE> __device__ __forceinline__ double lane_bcast(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), LANE), hi = __builtin_amdgcn_readlane((int)(b >> 32), LANE);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// Workgroup barrier for data that lives in LDS only.  __syncthreads() is fence + barrier: the fence makes every wave wait until ALL its
// outstanding memory operations have been acknowledged — global stores included (s_waitcnt vmcnt(0)), ≈ 1.3 µs for a wave that has just
// written node and item records (tools/level_prof.py: the selection → barrier phase of an item's first-tile workgroup took 2.0 – 2.3 µs
// against 0.72 for the others, and all four waves of that workgroup waited for it).  Where nothing another wave of the workgroup reads has
// gone through global memory, waiting for the LDS queue is enough: the stores drain behind the arithmetic that follows.
// BVH_LDS_BARRIER = 0 (developer variant): __syncthreads() everywhere.
#ifndef BVH_LDS_BARRIER
#define BVH_LDS_BARRIER 1
#endif
__device__ __forceinline__ void lds_barrier() {
#if BVH_LDS_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}
__device__ __forceinline__ unsigned long long lanemask_lt() {
    int l = lane_id();
    return l == 0 ? 0ull : (~0ull >> (64 - l));
}
__device__ __forceinline__ unsigned long long mask_range(int lo, int hi) {  // bits [lo, hi)
    unsigned long long h = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    unsigned long long l = lo >= 64 ? ~0ull : ((1ull << lo) - 1ull);
    return h & ~l;
}

// error plumbing -------------------------------------------------------------------------------
struct HipFail { hipError_t err; const char* what; int line; };
#define BVH_HIP(x)                                                     \
    do {                                                               \
        hipError_t _e = (x);                                           \
        if (_e != hipSuccess) throw ::bvhgpu::HipFail{_e, #x, __LINE__}; \
    } while (0)

}  // namespace bvhgpu
