"""include/gh_detlog.h: < 1 ulp against a 60-digit reference, and libm within 2 ulp of it."""
import decimal
import math
import random

import numpy as np

from oracle import c_oracle


def _ulp_err(y, exact):
    return abs(decimal.Decimal(y) - exact) / decimal.Decimal(math.ulp(float(exact)))


def test_detlog_accuracy():
    decimal.getcontext().prec = 60
    rnd = random.Random(7)
    xs = [rnd.uniform(1e-9, 1.0) for _ in range(4000)]
    xs += [(1 + rnd.randint(0, 500)) / (rnd.randint(1, 5) + rnd.randint(0, 5000)) for _ in range(4000)]
    xs += [10 ** rnd.uniform(-300, 300) for _ in range(2000)]
    xs += [1 + rnd.uniform(-1e-6, 1e-6) for _ in range(1000)]
    xs = np.array([x for x in xs if x != 1.0])
    det = np.zeros_like(xs)
    lm = np.zeros_like(xs)
    L = c_oracle.lib()
    L.orc_log10_many(xs.ctypes.data, det.ctypes.data, len(xs), 0)
    L.orc_log10_many(xs.ctypes.data, lm.ctypes.data, len(xs), 1)
    worst = 0.0
    for x, d in zip(xs, det):
        worst = max(worst, float(_ulp_err(d, decimal.Decimal(float(x)).log10())))
    assert worst < 1.0, worst
    # libm and the deterministic log10 never drift apart by more than 2 ulp
    assert np.all(np.abs(det - lm) <= 2 * np.spacing(np.abs(lm)))


def test_detlog_special_cases():
    L = c_oracle.lib()
    assert L.orc_log10(0.0, 0) == -math.inf
    assert L.orc_log10(math.inf, 0) == math.inf
    assert math.isnan(L.orc_log10(-1.0, 0))
    assert math.isnan(L.orc_log10(math.nan, 0))
    assert L.orc_log10(1.0, 0) == 0.0
    assert L.orc_log10(100.0, 0) == 2.0
    assert abs(L.orc_log10(5e-324, 0) - math.log10(5e-324)) < 1e-12
