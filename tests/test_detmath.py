"""include/rt_detmath.h: deterministic sin / cos / ln used by the sampling routines of both the
oracle and the f64 kernels.  Accuracy vs numpy (libm): within 2 ulp on the domains used."""
import numpy as np

from oracle import pyoracle


def ulp_err(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_sincos_accuracy_on_0_2pi():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(0, 2 * np.pi, 4000), [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi - 1e-12, 1e-300, 1e-9]])
    worst_s = worst_c = 0.0
    for x in xs:
        s, c, _ = pyoracle.detmath(x)
        ws, wc = np.sin(x), np.cos(x)
        # near a zero of sin/cos the absolute error is what matters (|err| <= 2^-53)
        es = min(ulp_err(s, ws), abs(s - ws) / 2.0 ** -53) if ws != 0 else abs(s)
        ec = min(ulp_err(c, wc), abs(c - wc) / 2.0 ** -53) if wc != 0 else abs(c)
        worst_s, worst_c = max(worst_s, es), max(worst_c, ec)
    assert worst_s <= 2.0 and worst_c <= 2.0, (worst_s, worst_c)
    assert tuple(pyoracle.detmath(0.0)[:2]) == (0.0, 1.0)


def test_log_accuracy_on_unit_interval():
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(0, 1, 3000), 10.0 ** rng.uniform(-16, 0, 1000), [1.0, 0.5, 2.0 ** -53, 0.7071, 0.70711, 1.4142, 3.7]])
    worst = 0.0
    for x in xs:
        got = pyoracle.detmath(x)[2]
        want = np.log(x)
        worst = max(worst, ulp_err(got, want) if want != 0 else abs(got))
    assert worst <= 2.0, worst
    assert pyoracle.detmath(1.0)[2] == 0.0


def test_inverse_trig_accuracy():
    """det_atan / det_atan2 / det_acos (UV maps of spheres and the sky, sphere.rs:69-75, sky.rs:40-46): within 2 ulp of libm
    on dense samples of their domains, exact on the special cases the kernels can meet (zeros, axis directions, +-1)."""
    rng = np.random.default_rng(3)
    worst_atan = worst_acos = worst_atan2 = 0.0
    xs = np.concatenate([rng.uniform(-4, 4, 4000), 10.0 ** rng.uniform(-12, 12, 1500), -(10.0 ** rng.uniform(-12, 12, 1500)),
                         [0.4375, 0.6875, 1.1875, 2.4375, 1.0, -1.0, 1e-300, 1e300]])
    for x in xs:
        got = pyoracle.detmath_inv(x)[0]
        worst_atan = max(worst_atan, ulp_err(got, np.arctan(x)))
    cs = np.concatenate([rng.uniform(-1, 1, 6000), 1.0 - 10.0 ** rng.uniform(-16, -0.3, 1500), -1.0 + 10.0 ** rng.uniform(-16, -0.3, 1500),
                         10.0 ** rng.uniform(-20, -1, 300), [0.5, -0.5, 0.0]])
    for c in cs:
        got = pyoracle.detmath_inv(c)[2]
        worst_acos = max(worst_acos, ulp_err(got, np.arccos(c)))
    ang = rng.uniform(-np.pi, np.pi, 6000)
    rad = 10.0 ** rng.uniform(-3, 3, 6000)
    for a, r in zip(ang, rad):
        y, x = r * np.sin(a), r * np.cos(a)
        got = pyoracle.detmath_inv(y, x)[1]
        want = np.arctan2(y, x)
        worst_atan2 = max(worst_atan2, ulp_err(got, want) if want != 0 else abs(got))
    assert worst_atan <= 2.0 and worst_acos <= 2.0 and worst_atan2 <= 2.0, (worst_atan, worst_acos, worst_atan2)
    inv = pyoracle.detmath_inv
    assert inv(1.0)[2] == 0.0 and inv(-1.0)[2] == np.pi and inv(0.0)[2] == np.pi / 2
    assert np.isnan(inv(1.0000001)[2]) and np.isnan(inv(float("nan"))[2])
    for y, x in [(0.0, 1.0), (-0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (1.0, -0.0), (0.0, 0.0), (-0.0, -0.0),
                 (float("inf"), 1.0), (1.0, float("inf")), (1.0, -float("inf")), (float("inf"), float("inf")), (-float("inf"), -float("inf")),
                 (1e-300, 1e300), (1e300, 1e-300), (1e-300, -1e300), (-1e300, -1e-300)]:
        got, want = inv(y, x)[1], np.arctan2(y, x)
        assert got == want and np.signbit(got) == np.signbit(want), (y, x, got, want)
