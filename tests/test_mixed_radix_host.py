"""Host-side facts the mixed-radix extension (gretel_amd/csrc/segmix.hpp) relies on, checked exhaustively on the CPU."""
import numpy as np


def test_float_quotient_of_the_state_walk_is_exact():
    # the oldest digit of a state is sigma // NI; the kernel takes it as trunc((float(sigma) + 0.5) * (1.0f / NI)) -- for every
    # NI up to the state budget and every sigma < 5 NI (the oldest digit's radix is at most 5)
    bad = 0
    for ni in range(1, 2049):
        sg = np.arange(0, 5 * ni, dtype=np.uint32)
        rcp = np.float32(1.0) / np.float32(ni)
        hi = ((sg.astype(np.float32) + np.float32(0.5)) * rcp).astype(np.uint32)
        bad += int((hi != sg // ni).sum())
        # ... and as the kernel writes it, one fused multiply-add: fma(float(sigma), rcp, 0.5f * rcp) (evaluated here in binary64 and
        # rounded once to binary32, which is what a binary32 fma of these operands gives: products of two floats are exact doubles)
        f = (sg.astype(np.float64) * np.float64(rcp) + np.float64(np.float32(0.5) * rcp)).astype(np.float32)
        bad += int((f.astype(np.uint32) != sg // ni).sum())
    assert bad == 0


def test_reciprocal_multiplies_of_the_task_decode_are_exact():
    # young digits: j // r for r in 1..5 and j < 125 (three digits of radix <= 5) as (j * ceil(32768 / r)) >> 15
    for r in range(1, 6):
        j = np.arange(0, 2048, dtype=np.uint64)
        assert np.array_equal((j * ((32768 + r - 1) // r)) >> 15, j // r)


def test_mixed_geometry_matches_the_ranked_one():
    # class 6 cuts a window into the segments and groups of the ranked layout (the host sizes grids by the larger of the classes)
    import math

    def geom(n, ns, g1max):
        g2 = min(16, max(1, (65536 if ns > 3125 else 32768) // ns))
        ln = max(8, math.ceil(n / (g1max * g2)))
        return ln, math.ceil(n / ln), g2
    for n in (5, 64, 700, 10000, 20011, 50000):
        assert geom(n, 1024, 16) == geom(n, 2048, 16)
