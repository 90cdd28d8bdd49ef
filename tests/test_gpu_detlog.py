"""The DEVICE build of include/gh_detlog.h against the host's libm log10 (the reference's: math.log10,
gretel/gretel.py:2): bit for bit on 2e7 arguments, through the C ABI (gh_log10_device)."""
import math

import numpy as np
import pytest

from gretel_amd import _lib
from test_detlog import arguments, libm_log10

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [3, 4])
def test_device_log10_is_libm_bit_for_bit(seed):
    x = arguments(3_000_000, seed)
    ref = libm_log10(x)
    y = _lib.log10_many(x, device=0)
    bad = np.flatnonzero(y.view(np.int64) != ref.view(np.int64))
    assert bad.size == 0, "%d of %d differ from libm, first x=%s libm=%s device=%s" % (
        bad.size, x.size, float(x[bad[0]]).hex(), float(ref[bad[0]]).hex(), float(y[bad[0]]).hex())


def test_device_special_cases():
    x = np.array([0.0, -0.0, math.inf, -1.0, math.nan, 1.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 100.0])
    y = _lib.log10_many(x, device=0)
    assert y[0] == -math.inf and y[1] == -math.inf and y[2] == math.inf and math.isnan(y[3]) and math.isnan(y[4])
    assert y[5] == 0.0 and math.copysign(1.0, y[5]) == 1.0
    assert y[6:].tolist() == [math.log10(v) for v in x[6:].tolist()]
