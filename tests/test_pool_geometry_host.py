"""Host-side facts about how the candidate-pool extension (gretel_amd/csrc/cwalk.hpp; gretel/gretel.py:143-189 for 6 <= L <= 128)
cuts a window, through the C ABI (gh_debug_pool_geometry needs no GPU): the LDS a walker is launched with fits a CU for every lag
count, a packed walker's chunk is at least one unrolled block of L steps, and every position belongs to exactly one segment."""
import ctypes as C

import numpy as np
import pytest

from gretel_amd import _lib

LDS_PER_CU = 160 * 1024
STATIC_PACKED = 1024            # k_cwalk besides its dynamic slice: the staged keys of the next pool, two counters
STATIC_BYTES = 9 * 1024         # k_cwalkg: the rings of picks (64 entries x 128 bytes), counters


def geometry(n, L, five):
    out = (C.c_int64 * 6)()
    _lib.check(_lib.load().gh_debug_pool_geometry(n, L, 1 if five else 0, out))
    return [int(x) for x in out]


@pytest.mark.parametrize("five", [False, True])
def test_every_lag_count_fits_the_lds_and_keeps_its_unrolled_block(five):
    for L in range(6, 129):
        S, seglen, chunk, lds, packed, threads = geometry(10000, L, five)
        assert packed == (1 if L <= (21 if five else 32) else 0)
        assert lds + (STATIC_PACKED if packed else STATIC_BYTES) <= LDS_PER_CU, (L, lds)
        assert chunk >= 1
        if packed:
            assert chunk >= L, (L, chunk)        # the walk unrolled over L steps runs at least once per chunk
            assert chunk <= 64
        rows_cols = 25 if five else 16
        assert lds == chunk * (L * rows_cols + 5) * 8
        assert threads == (512 if five else 256)


def test_two_walkers_share_a_cu_up_to_thirteen_lags():
    for L in range(6, 14):
        assert 2 * (geometry(50000, L, False)[3] + STATIC_PACKED) <= LDS_PER_CU
    for L in range(6, 13):
        assert 2 * (geometry(50000, L, True)[3] + STATIC_PACKED) <= LDS_PER_CU


@pytest.mark.parametrize("n", [1, 31, 32, 33, 700, 10000, 16385, 50000, 200000])
def test_segments_cover_the_window_once(n):
    for L in (6, 13, 14, 32, 33, 128):
        S, seglen = geometry(n, L, False)[:2]
        assert S >= 1 and seglen >= 32
        assert (S - 1) * seglen < n <= S * seglen
        assert S <= (512 if L <= 13 else 256)


def test_arguments_are_checked():
    out = (C.c_int64 * 6)()
    lib = _lib.load()
    assert lib.gh_debug_pool_geometry(100, 5, 0, out) != 0
    assert lib.gh_debug_pool_geometry(100, 129, 0, out) != 0
    assert lib.gh_debug_pool_geometry(0, 8, 0, out) != 0
    assert lib.gh_debug_pool_geometry(100, 8, 0, None) != 0
