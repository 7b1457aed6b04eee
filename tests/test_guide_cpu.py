"""The containment argument of the f64 guide walk (bvh_amd/csrc/common.hpp "guide boxes", traverse.hip "guide walk"), replayed in numpy:
an f64 box grown by 2^-18 x S and rounded outward to f32, tested with the round-to-nearest f32 copy of an f64 ray whose origin lies
within 3 x S and whose |1/d| x 4 S lies inside 2^+-100, must pass the f32 slab test whenever the f64 box passes the f64 test — on rays
aimed AT the faces, edges and corners of the boxes (grazing cases, where a rounding could flip the outcome), flat boxes and rays that
are nearly parallel to an axis."""
import numpy as np

GROW, ORIGIN_MAX = 2.0 ** -18, 3.0


def _exact64(o, inv, mn, mx):
    tl, th = (mn - o) * inv, (mx - o) * inv
    tmin, tmax = np.minimum(tl, th).max(axis=1), np.maximum(tl, th).min(axis=1)
    return (tmax >= tmin) & (tmax >= 0)


def _below(x):   # largest float32 <= x
    f = x.astype(np.float32)
    return np.where(f.astype(np.float64) > x, np.nextafter(f, np.float32(-np.inf)), f)


def _guide32(o, inv, mn, mx, S):
    mn32, mx32 = _below(mn - GROW * S), -_below(-(mx + GROW * S))
    o32, inv32 = o.astype(np.float32), inv.astype(np.float32)
    tl, th = (mn32 - o32) * inv32, (mx32 - o32) * inv32          # float32 arithmetic, round to nearest
    tmin, tmax = np.minimum(tl, th).max(axis=1), np.maximum(tl, th).min(axis=1)
    return (tmax >= tmin) & (tmax >= 0)


def _in_range(o, inv, S):
    ai = np.abs(inv) * (4.0 * S)
    return np.all((np.abs(o) <= ORIGIN_MAX * S) & (ai <= 2.0 ** 100) & (ai >= 2.0 ** -100), axis=1)


def test_guide_box_test_never_rejects_what_the_f64_test_accepts():
    rng = np.random.default_rng(11)
    N, S = 400_000, 1000.0
    exact_hits = 0
    for trial in range(12):
        c = rng.uniform(-S, S, (N, 3))
        h = np.abs(rng.normal(0, [0.5, 2.0, 50.0][trial % 3], (N, 3)))
        if trial % 4 == 3:
            h[:, rng.integers(0, 3)] = 0.0                       # flat boxes
        mn, mx = np.maximum(c - h, -S), np.minimum(c + h, S)    # (the scene's largest |coordinate| is S)
        o = rng.uniform(-ORIGIN_MAX * S, ORIGIN_MAX * S, (N, 3))
        tgt = c + rng.choice([-1.0, 1.0], (N, 3)) * h * (1 + rng.normal(0, 1e-7, (N, 3)) * (trial % 2))
        d = tgt - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        if trial >= 6:                                           # nearly parallel to an axis
            k = rng.integers(0, 3, N)
            d[np.arange(N), k] *= 10.0 ** rng.uniform(-12, -3, N)
        inv = 1.0 / d
        ok = _in_range(o, inv, S)
        e, g = _exact64(o, inv, mn, mx), _guide32(o, inv, mn, mx, S)
        assert not np.any(e & ~g & ok), f"trial {trial}: the guide test rejected {int((e & ~g & ok).sum())} boxes the f64 test accepts"
        exact_hits += int((e & ok).sum())
    assert exact_hits > 1_000_000                                # (the cases are grazing hits, not misses)


def test_guide_range_excludes_what_the_argument_does_not_cover():
    S = 10.0
    o = np.array([[0.0, 0.0, 0.0], [31.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    d = np.array([[1.0, 1.0, 1.0], [1.0, 1.0, 1.0], [1.0, 0.0, 1.0], [1.0, 1e-140, 1.0]])
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    assert _in_range(o, inv, S).tolist() == [True, False, False, False]   # far origin, axis-parallel (1/0 = inf), 1/d beyond 2^100 / (4 S)


# ---- the bound with every rounding ADVERSARIAL ("no machine-checked worst case") -----------------------------------------
# Round-to-nearest lands between the two directed roundings of every step, so if the guide test still accepts when each of its five
# roundings per plane (box bound outward as the engine rounds it; origin, 1/d, difference, product: whichever direction hurts) is pushed
# a whole ulp the wrong way, it accepts under round-to-nearest too.  Priced: 30 x 2^-24 S against the growth's 64 x 2^-24 S.
F32_MAX, F32_MIN_NORMAL, SCENE_MIN, SCENE_MAX = float(np.finfo(np.float32).max), 2.0 ** -126, 2.0 ** -125, 2.0 ** 125


def _above(x):
    return -_below(-x)


def _in_range_v2(o, inv, S):
    """guide_ray_load's range test as of round 4 (scale-relative AND absolute conditions)"""
    ai = np.abs(inv) * (4.0 * S)
    per_axis = ((np.abs(o) <= ORIGIN_MAX * S) & (np.abs(o) <= F32_MAX) & (ai <= 2.0 ** 100) & (ai >= 2.0 ** -100) &
                (np.abs(inv) <= F32_MAX) & (np.abs(inv) >= F32_MIN_NORMAL))
    return np.all(per_axis, axis=1) & (S >= SCENE_MIN) & (S <= SCENE_MAX)


def _guide32_adversarial(o, inv, mn, mx, S):
    """the f32 guide test with directed roundings chosen per plane to make it REJECT: near-plane t as large, far-plane t as small as
    any rounding of (b32 - o32) * inv32 can make them"""
    with np.errstate(over="ignore", invalid="ignore", under="ignore"):
        mn32, mx32 = _below(mn - GROW * S).astype(np.float64), _above(mx + GROW * S).astype(np.float64)
        o_c = [_below(o).astype(np.float64), _above(o).astype(np.float64)]
        i_c = [_below(inv).astype(np.float64), _above(inv).astype(np.float64)]
        near = np.full(o.shape, -np.inf)
        far = np.full(o.shape, np.inf)
        for b32, which in ((mn32, 0), (mx32, 1)):
            lo, hi = np.full(o.shape, np.inf), np.full(o.shape, -np.inf)
            for oc in o_c:
                for d in (_below(b32 - oc).astype(np.float64), _above(b32 - oc).astype(np.float64)):
                    for ic in i_c:
                        p = d * ic                                # exact in f64: two 24-bit significands
                        lo = np.minimum(lo, _below(p).astype(np.float64))
                        hi = np.maximum(hi, _above(p).astype(np.float64))
            # plane `which` is the near plane of its axis when (b - o) * inv is the smaller of the two products
            if which == 0:
                t_mn_lo, t_mn_hi = lo, hi
            else:
                t_mx_lo, t_mx_hi = lo, hi
        pos = inv > 0
        near = np.where(pos, t_mn_hi, t_mx_hi)                   # the entering plane, pushed up
        far = np.where(pos, t_mx_lo, t_mn_lo)                    # the leaving plane, pushed down
        tmin, tmax = near.max(axis=1), far.min(axis=1)
        return (tmax >= tmin) & (tmax >= 0)


def _edge_cases(rng, N, S, o_scale, inv_mag):
    """grazing rays at the edges of the range: |o| up to exactly 3 S, |1/d| at the given magnitudes (raw rays: the C ABI takes any inv)"""
    c = rng.uniform(-S, S, (N, 3))
    h = np.abs(rng.normal(0, S * rng.choice([1e-6, 1e-3, 0.05], (N, 1)), (N, 3)))
    h[rng.random(N) < 0.2, rng.integers(0, 3)] = 0.0
    mn, mx = np.maximum(c - h, -S), np.minimum(c + h, S)
    o = rng.uniform(-o_scale * S, o_scale * S, (N, 3))
    edge = rng.random((N, 3)) < 0.3
    o = np.where(edge, np.sign(o) * ORIGIN_MAX * S, o)            # exactly on the origin limit
    corner = np.where(rng.random((N, 3)) < 0.5, mn, mx)
    tgt = corner * (1 + rng.normal(0, 2.0 ** -50, (N, 3)))        # the ray grazes a corner / edge of the box
    d = tgt - o
    nrm = np.abs(d).max(axis=1, keepdims=True)
    d = d / np.where(nrm > 0, nrm, 1.0)
    with np.errstate(divide="ignore", over="ignore"):
        inv = 1.0 / d
        inv = inv * inv_mag                                       # (an unnormalised direction scales every t alike: the hit set is the same)
    return o, inv, mn, mx


def test_guide_bound_with_directed_roundings_at_the_range_edges():
    rng = np.random.default_rng(2024)
    N = 60_000
    accepted = 0
    # (S, origin scale, |1/d| scale): ordinary scenes, S at the smallest / largest the range admits, 4 S |1/d| at 2^+-100
    grid = [(1000.0, 3.0, 1.0), (1.0, 3.0, 1.0), (2.0 ** -125, 3.0, 2.0 ** 20), (2.0 ** -124, 1.0, 2.0 ** 100), (2.0 ** -60, 3.0, 2.0 ** 150),
            (2.0 ** 60, 3.0, 2.0 ** -150), (2.0 ** 96, 3.0, 1.0), (2.0 ** 125, 3.0, 2.0 ** -30), (F32_MAX / 4, 3.0, 2.0 ** -30), (F32_MAX * 0.999, 1.0, 2.0 ** -60),
            (2.0 ** -100, 3.0, 2.0 ** -2), (2.0 ** 40, 2.9999, 2.0 ** 58), (2.0 ** -140, 3.0, 2.0 ** 40), (F32_MAX * 2, 1.0, 2.0 ** -40)]
    for S, o_scale, inv_mag in grid:
        o, inv, mn, mx = _edge_cases(rng, N, S, o_scale, inv_mag)
        with np.errstate(over="ignore", invalid="ignore", under="ignore"):
            ok = _in_range_v2(o, inv, S)
            e = _exact64(o, inv, mn, mx)
            g = _guide32_adversarial(o, inv, mn, mx, S)
        bad = e & ok & ~g
        assert not np.any(bad), (f"S = {S:g}, |1/d| x {inv_mag:g}: {int(bad.sum())} boxes the f64 test accepts are rejected by the guide test "
                                 f"under adversarial roundings although the ray is in range, e.g. o = {o[bad][0]}, inv = {inv[bad][0]}")
        accepted += int((e & ok).sum())
        if S < SCENE_MIN or S > SCENE_MAX:
            assert not ok.any()                                   # such scenes never take the guide walk
    assert accepted > 100_000                                     # the in-range cases really are (grazing) hits


def test_round4_range_test_closes_the_absolute_gaps():
    """the scale-relative test |1/d| * 4S in 2^+-100 alone lets (float)(1/d) overflow on a very small scene (or underflow on a
    huge one) unflagged, and a denormal-sized scene's coordinates round by more than the growth covers."""
    one = np.ones((1, 3))
    assert _in_range(one * 0.0, one * 2.0 ** 130, 2.0 ** -60)[0] and not _in_range_v2(one * 0.0, one * 2.0 ** 130, 2.0 ** -60)[0]     # inv32 = inf
    assert _in_range(one * 0.0, one * 2.0 ** -140, 2.0 ** 100)[0] and not _in_range_v2(one * 0.0, one * 2.0 ** -140, 2.0 ** 100)[0]   # inv32 denormal
    assert _in_range(one * 0.0, one * 2.0 ** 40, 2.0 ** -140)[0] and not _in_range_v2(one * 0.0, one * 2.0 ** 40, 2.0 ** -140)[0]     # denormal scene
    assert _in_range_v2(one * 3.0, one * -1.5, 1.0)[0]
