"""include/crh_fmath.h: the deterministic elementary functions that replace libm on the tessellation path."""
import numpy as np
import pytest

pytestmark = pytest.mark.usefixtures("oracle_lib")


def ulp_error(got, exact):
    exact32 = exact.astype(np.float32)
    ulp = np.abs(np.nextafter(exact32, np.float32(np.inf)) - exact32).astype(np.float64)
    return np.abs(got.astype(np.float64) - exact) / np.maximum(ulp, 1e-300)


def test_within_one_ulp_of_libm():
    from oracle.binding import fmath_eval
    rng = np.random.RandomState(0)
    n = 200000
    x = (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-4, 3, n)).astype(np.float32)
    y = (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-4, 3, n)).astype(np.float32)
    assert ulp_error(fmath_eval(0, y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() <= 0.5000001
    c = np.concatenate([rng.uniform(-1, 1, n), 1 - 10.0 ** rng.uniform(-7, -1, n), -1 + 10.0 ** rng.uniform(-7, -1, n)]).astype(np.float32)
    assert ulp_error(fmath_eval(1, c), np.arccos(c.astype(np.float64))).max() <= 0.5000001
    a = rng.uniform(-8, 8, n).astype(np.float32)
    assert ulp_error(fmath_eval(2, a), np.sin(a.astype(np.float64))).max() <= 0.5000001
    assert ulp_error(fmath_eval(3, a), np.cos(a.astype(np.float64))).max() <= 0.5000001
    b = np.concatenate([rng.uniform(1e-6, 4, n), 1 + rng.uniform(-1e-5, 1e-5, n)]).astype(np.float32)
    e = np.concatenate([rng.uniform(-3, 3, n), 1.0 / rng.randint(1, 65, n)]).astype(np.float32)
    assert ulp_error(fmath_eval(4, b, e), np.power(b.astype(np.float64), e.astype(np.float64))).max() <= 0.5000001


def test_special_values():
    from oracle.binding import fmath_eval
    f = np.float32
    assert fmath_eval(0, [f(0.0)], [f(-1.0)])[0] == f(np.pi) and fmath_eval(0, [f(-0.0)], [f(-1.0)])[0] == f(-np.pi)
    assert fmath_eval(0, [f(0.0)], [f(0.0)])[0] == 0.0
    assert np.isnan(fmath_eval(1, [f(1.0000001)])[0]) and np.isnan(fmath_eval(1, [f(-1.0000001)])[0])
    assert fmath_eval(1, [f(1.0)])[0] == 0.0 and fmath_eval(1, [f(-1.0)])[0] == f(np.pi)
    assert fmath_eval(5, [f(-3.5)], [f(2.0)])[0] == f(-1.5)  # WGSL %: x - y*trunc(x/y), sign of the dividend
    assert fmath_eval(5, [f(7.25)], [f(2.0)])[0] == f(1.25)


def test_polynomial_solvers_find_the_roots():
    """The un-vendored geometric_algebra::polynomial solvers are restated as textbook closed forms; check residuals."""
    from oracle.binding import solve
    rng = np.random.RandomState(1)
    for degree in (1, 2, 3, 4):
        for _ in range(300):
            true_roots = rng.uniform(-2, 2, degree)
            coefficients = np.poly(true_roots)[::-1] * rng.uniform(0.5, 2.0)  # ascending
            disc, roots = solve(degree, coefficients)
            assert len(roots) == degree
            t = roots[:, 0].astype(np.float64) / roots[:, 2]
            assert np.abs(roots[:, 1]).max() < 2e-2  # real roots (tiny imaginary parts only from near-double roots)
            assert np.allclose(np.sort(t), np.sort(true_roots), atol=2e-2)
    # complex pair: x^2 + 1 -> real part 0, imaginary +-1 ; cubic with one real root reports it at index 0
    disc, roots = solve(2, [1.0, 0.0, 1.0])
    assert disc < 0 and roots[:, 0].tolist() == [0.0, 0.0] and sorted(roots[:, 1].tolist()) == [-2.0, 2.0]
    disc, roots = solve(3, [-1.0, 0.0, 0.0, 1.0])  # t^3 - 1
    assert disc < 0 and abs(roots[0, 0] / roots[0, 2] - 1.0) < 1e-6 and roots[0, 1] == 0.0
    disc, roots = solve(3, [0.0, -1.0, 0.0, 1.0])  # t^3 - t: three real roots, positive discriminant
    assert disc > 0 and np.allclose(sorted(roots[:, 0] / roots[:, 2]), [-1, 0, 1], atol=1e-6)
    # degree drop below the error margin (curve.rs passes ERROR_MARGIN)
    disc, roots = solve(2, [1.0, -2.0, 1e-5])
    assert len(roots) == 1 and abs(roots[0, 0] / roots[0, 2] - 0.5) < 1e-6
