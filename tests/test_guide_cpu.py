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
