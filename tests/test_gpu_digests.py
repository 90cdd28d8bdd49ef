"""Full-depth parity against committed digests (tests/golden/fullsize_digests.json, written by the C oracle on libm in the
build container: tests/golden/make_fullsize_digests.py).  Every case goes through the HIP path (C ABI) at its full depth --
C5 x 1000 paths, C5 x 100 under C and E + marginal term + f64, C3 seeds 0..7 x 100 (C4's windows), C3 x 100 under the 20
specs of bench.py's spec_matrix, the sparse-deletion window x 100 -- and is compared path by path: the sha256 of the path
bytes, hp_current / hp_original / ratio bit for bit, the removed mass to 1e-10 relative, and the sha256 of the reweighted
tensor behind the last path.  The oracle is not needed on the GPU box for these (reference: gretel/gretel.py:79-98,143-189,
gretel/cmd.py:148-179)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import make_fullsize_digests as mk  # noqa: E402

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "fullsize_digests.json")) as _f:
    DIGESTS = json.load(_f)["cases"]

_tables = {}


def _table(case):
    key = (case["config"], case["seed"], case.get("table"))
    if key not in _tables:
        if len(_tables) >= 2:           # C3 tables are 30 MB, C5 larger: keep two
            _tables.pop(next(iter(_tables)))
        _tables[key] = mk.make_table(case)
    return _tables[key]


def _ordered():
    # cases that share a table next to each other
    return sorted(DIGESTS, key=lambda k: (DIGESTS[k]["case"]["config"], DIGESTS[k]["case"]["seed"], str(DIGESTS[k]["case"].get("table")), k))


@pytest.mark.parametrize("name", _ordered())
def test_full_depth_against_the_oracles_digests(name):
    from gretel_amd.hansel import Hansel
    want = DIGESTS[name]
    case = want["case"]
    t = _table(case)
    assert (t.n_snps, t.band) == (want["n_snps"], want["band"])
    h = Hansel(t.n_snps, band=t.band, **case["spec"])
    st = h.fill_from_support(t.rank, t.off, t.bases)
    assert [int(x) for x in st] == want["fill_stats"] and h.L == want["L"]
    res = h.spin(case["paths"])
    got = mk.digest_result(res, h.export_band())
    assert got["n"] == want["n"] == case["paths"] and got["hole_at"] == want["hole_at"]
    bad = [q for q in range(got["n"]) if got["path_sha"][q] != want["path_sha"][q]]
    assert not bad, "recovered SNP sequences differ from the oracle's, first at path %d of %d" % (bad[0], got["n"])
    for key in ("hp_current", "hp_original", "ratio"):
        bad = [q for q in range(got["n"]) if got[key][q] != want[key][q]]
        assert not bad, "%s differs, first at path %d: %s vs %s" % (key, bad[0], got[key][bad[0]], want[key][bad[0]])
    mg = np.array([float.fromhex(x) for x in got["magnitude"]]), np.array([float.fromhex(x) for x in want["magnitude"]])
    assert np.allclose(mg[0], mg[1], rtol=1e-10, atol=0)
    assert got["band_sha"] == want["band_sha"], "the reweighted tensor behind the last path differs"


def test_the_digest_file_covers_what_it_says():
    names = set(DIGESTS)
    assert "C5/seed0/default/1000" in names and DIGESTS["C5/seed0/default/1000"]["n"] == 1000
    assert all("C3/seed%d/default/100" % s in names for s in range(8))
    assert sum(1 for k in names if k.startswith("C3/seed0/cond_mode=")) == 19      # + the default spec = spec_matrix's 20
    assert "C3/seed0/sparse_deletions/default/100" in names
    assert {c[0] for c in mk.cases()} == names
