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
