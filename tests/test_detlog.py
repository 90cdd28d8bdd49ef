"""include/gh_detlog.h IS the running libm's log10, bit for bit (it restates glibc's; the reference takes its log10
from libm through math.log10, gretel/gretel.py:2).  Two builds of the same source are held to libm here on the CPU:
the C oracle's (gcc) and the product library's host-side entry gh_log10_host (hipcc's host pass); the device build
is held to libm in tests/test_gpu_detlog.py.  numpy's log10 is NOT the yardstick -- it may dispatch to its own SIMD
routines; libm's log10 is reached through the C oracle (orc_log10_many(..., use_libm=1)) and through math.log10."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from gretel_amd import _lib
from oracle import c_oracle


def arguments(n, seed):
    """What the hot path takes logs of, and what could break a restatement: every binade, the near-one interval
    glibc treats separately, the edges of its 128 table intervals, subnormals, and quotients (1+H)/(V+T) of counts
    that reweights have made fractional."""
    rng = np.random.default_rng(seed)
    parts = []
    bits = rng.integers(0, 0x7ff0000000000000, size=n, dtype=np.int64)          # every positive finite double, uniformly in bits
    parts.append(bits.view(np.float64))
    parts.append(rng.random(n) * 2.0)                                            # (0, 2): marginals, conditionals
    parts.append(1.0 + (rng.random(n) - 0.5) * 0.14)                             # around glibc's near-one interval [0.9375, 1.0647)
    edges = (0x3fe6000000000000 + (np.arange(0, 256, dtype=np.int64) << 45))     # table interval edges over two binades
    e = (edges[None, :] + rng.integers(-3, 4, size=(max(1, n // 256), 256))).ravel()
    parts.append(e.view(np.float64))
    h = np.floor(rng.random(n) * 5000) * np.where(rng.random(n) < 0.5, 1.0, rng.random(n))      # counts, some reweighted
    t = h + np.floor(rng.random(n) * 800) * np.where(rng.random(n) < 0.5, 1.0, rng.random(n))
    v = rng.integers(1, 6, size=n)
    for dt in (np.float32, np.float64):
        parts.append((1.0 + h.astype(dt).astype(np.float64)) / (v + t.astype(dt).astype(np.float64)))
    parts.append(np.ldexp(rng.random(n // 8 + 1), rng.integers(-1074, -1000, size=n // 8 + 1)))  # subnormals and their neighbours
    x = np.concatenate(parts)
    return x[(x > 0) & np.isfinite(x)]


def libm_log10(x):
    y = np.empty_like(x)
    c_oracle.lib().orc_log10_many(x.ctypes.data, y.ctypes.data, x.size, 1)
    return y


def oracle_build(x):
    y = np.empty_like(x)
    c_oracle.lib().orc_log10_many(x.ctypes.data, y.ctypes.data, x.size, 0)
    return y


def _other_libm_reason():
    """include/gh_detlog.h restates ONE libm: glibc >= 2.28's table-driven log in the form hosts with FMA run it (__log_fma).
    On a host whose libm takes another road -- no FMA, an older glibc, another libc -- the comparison below cannot hold and
    says nothing about the header: skip, with the reason.  (None: this host runs the build the header restates.)"""
    import platform
    libc, ver = platform.libc_ver()
    if libc != "glibc":
        return "the C library is %r, not glibc" % (libc or "unknown")
    try:
        if tuple(int(q) for q in ver.split(".")[:2]) < (2, 28):
            return "glibc %s predates the table-driven log (2.28)" % ver
    except ValueError:
        return "cannot read the glibc version %r" % ver
    if platform.machine() not in ("x86_64", "AMD64"):
        return "machine %s: the restated instruction sequence is the x86-64 one" % platform.machine()
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
    except Exception:
        return None
    if "fma" not in flags or "avx2" not in flags:
        return "this CPU has no FMA/AVX2: glibc's ifunc picks the non-FMA log here"
    return None


def test_the_table_is_what_its_construction_says():
    # tools/gen_logtab.py --verify: c at the centre of its subinterval, log c = round(2^43 ln(1/invc)) / 2^43 (exact arithmetic)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_logtab
    assert gen_logtab.verify() == []


@pytest.mark.parametrize("seed", [1, 2])
def test_restated_log10_is_libm_bit_for_bit(seed):
    reason = _other_libm_reason()
    if reason:
        pytest.skip("include/gh_detlog.h restates glibc >= 2.28's __log_fma; " + reason)
    x = arguments(3_000_000, seed)
    assert x.size > 15_000_000
    ref = libm_log10(x)
    for name, y in (("gcc build (oracle)", oracle_build(x)), ("hipcc host build (product)", _lib.log10_many(x))):
        bad = np.flatnonzero(y.view(np.int64) != ref.view(np.int64))
        assert bad.size == 0, "%s: %d of %d differ from libm, first x=%s libm=%s here=%s" % (
            name, bad.size, x.size, float(x[bad[0]]).hex(), float(ref[bad[0]]).hex(), float(y[bad[0]]).hex())
    # and libm through the C oracle is libm through Python (the road the reference takes)
    k = np.random.default_rng(seed).integers(0, x.size, size=20000)
    assert [math.log10(v) for v in x[k].tolist()] == ref[k].tolist()


def test_special_cases():
    L = c_oracle.lib()
    for libm in (0, 1):
        assert L.orc_log10(0.0, libm) == -math.inf
        assert L.orc_log10(math.inf, libm) == math.inf
        assert math.isnan(L.orc_log10(-1.0, libm))
        assert math.isnan(L.orc_log10(math.nan, libm))
        assert L.orc_log10(1.0, libm) == 0.0 and math.copysign(1.0, L.orc_log10(1.0, libm)) == 1.0
        assert L.orc_log10(100.0, libm) == 2.0
    assert L.orc_log10(5e-324, 0) == math.log10(5e-324)
    x = np.array([0.0, -0.0, math.inf, -1.0, math.nan, 1.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308])
    y = _lib.log10_many(x)
    assert y[0] == -math.inf and y[1] == -math.inf and y[2] == math.inf and math.isnan(y[3]) and math.isnan(y[4]) and y[5] == 0.0
    assert y[6:].tolist() == [math.log10(v) for v in x[6:].tolist()]


def test_committed_table_is_the_running_libms():
    # include/gh_logtab.inc was read out of a glibc 2.35 libm; on a host with another glibc >= 2.28 it must still be the same table
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_logtab.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
