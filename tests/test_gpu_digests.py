"""Full-depth parity against committed digests (tests/golden/fullsize_digests.json, written by the C oracle on libm in the
build container: tests/golden/make_fullsize_digests.py).  Every case goes through the HIP path (C ABI) at its full depth --
C5 x 1000 paths, C5 x 100 under C and E + marginal term + f64, C3 seeds 0..7 x 100 (C4's windows), C3 x 100 under the 20
specs of bench.py's spec_matrix, the sparse-deletion window x 100; round 6: deletion columns at L = 3..22, dense deletions at
L = 4..12, states as bytes at L = 33, 40, 48 -- and is compared path by path: the sha256 of the path
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
    if case.get("L"):
        assert [int(x) for x in st] == want["fill_stats"]
        h.L = case["L"]             # (the lag count of the case, not the fill's: gretel/util.py:333 gives one per window)
    assert [int(x) for x in st] == want["fill_stats"] and h.L == want["L"]
    res = h.spin(case["paths"])
    got = mk.digest_result(res, h.export_band())
    assert got["n"] == case["paths"]
    _compare(got, want, name)

def _compare(got, want, what):
    assert got["n"] == want["n"] and got["hole_at"] == want["hole_at"], what
    bad = [q for q in range(got["n"]) if got["path_sha"][q] != want["path_sha"][q]]
    assert not bad, "%s: recovered SNP sequences differ from the oracle's, first at path %d of %d" % (what, bad[0], got["n"])
    for key in ("hp_current", "hp_original", "ratio"):
        bad = [q for q in range(got["n"]) if got[key][q] != want[key][q]]
        assert not bad, "%s: %s differs, first at path %d: %s vs %s" % (what, key, bad[0], got[key][bad[0]], want[key][bad[0]])
    mg = np.array([float.fromhex(x) for x in got["magnitude"]]), np.array([float.fromhex(x) for x in want["magnitude"]])
    assert np.allclose(mg[0], mg[1], rtol=1e-10, atol=0), what
    assert got["band_sha"] == want["band_sha"], "%s: the reweighted tensor behind the last path differs" % what


def test_the_window_pipeline_at_full_depth_c4s_eight_windows_in_one_batch(monkeypatch):
    """Throughput mode (gretel_amd/csrc/wpipe.hpp) against the same digests: the eight 10k-SNP windows of C4 as ONE batch, 100
    paths each, every window carried through all its paths by its persistent workgroup."""
    from gretel_amd.hansel import Hansel, HanselBatch
    monkeypatch.setenv("GH_PIPE_MIN", "1")
    names = ["C3/seed%d/default/100" % s for s in range(8)]
    hs = []
    for nm in names:
        case = DIGESTS[nm]["case"]
        t = mk.make_table(case)
        h = Hansel(t.n_snps, band=t.band, **case["spec"])
        st = h.fill_from_support(t.rank, t.off, t.bases)
        assert [int(x) for x in st] == DIGESTS[nm]["fill_stats"]
        hs.append(h)
        del t
    b = HanselBatch(hs)
    res = b.spin(100)
    info = b.pipe_info()
    assert info["windows"] == 8 and info["handed_back"] == 0, info
    for nm, h, r in zip(names, hs, res):
        _compare(mk.digest_result(r, h.export_band()), DIGESTS[nm], nm + " through the pipeline")


@pytest.mark.parametrize("storage", ["f32", "f64"])
def test_the_window_pipeline_at_full_depth_under_every_spec(storage, monkeypatch):
    """C3 seed 0 x 100 paths under the specs of bench.py's spec_matrix, each as a batch of one window through the pipeline."""
    from gretel_amd.hansel import Hansel, HanselBatch
    monkeypatch.setenv("GH_PIPE_MIN", "1")
    names = sorted(k for k in DIGESTS if k.startswith("C3/seed0/cond_mode=") and k.endswith("storage=%s/100" % storage))
    if storage == "f32":
        names.append("C3/seed0/default/100")
    assert len(names) == 10
    t = mk.make_table(DIGESTS[names[0]]["case"])
    for nm in names:
        case = DIGESTS[nm]["case"]
        h = Hansel(t.n_snps, band=t.band, **case["spec"])
        h.fill_from_support(t.rank, t.off, t.bases)
        b = HanselBatch([h])
        r = b.spin(100)[0]
        assert b.pipe_info()["windows"] == 1, (nm, b.pipe_info())
        _compare(mk.digest_result(r, h.export_band()), DIGESTS[nm], nm + " through the pipeline")
        del b, h


@pytest.mark.parametrize("name", ["C5/seed0/default/1000", "C5/seed0/cond_mode=C/100", "C5/seed0/cond_mode=E-marginal_term=True-storage=f64/100"])
def test_the_window_pipeline_at_full_depth_c5(name, monkeypatch):
    """C5 (50k SNPs, L = 11, the deep spin of 1 000 paths) as a batch of one window through the pipeline -- which by default
    leaves lag counts beyond ten to the candidate pools (GH_PIPE_MAX_L): a second, independent HIP flow against the same digest."""
    from gretel_amd.hansel import Hansel, HanselBatch
    monkeypatch.setenv("GH_PIPE_MIN", "1")
    monkeypatch.setenv("GH_PIPE_MAX_L", "14")
    want = DIGESTS[name]
    case = want["case"]
    t = _table(case)
    h = Hansel(t.n_snps, band=t.band, **case["spec"])
    h.fill_from_support(t.rank, t.off, t.bases)
    assert h.L == want["L"] == 11
    b = HanselBatch([h])
    r = b.spin(case["paths"])[0]
    info = b.pipe_info()
    assert info["windows"] == 1 and info["handed_back"] == 0 and info["threads"] == 512, info
    _compare(mk.digest_result(r, h.export_band()), want, name + " through the pipeline")


@pytest.mark.parametrize("name", ["C3/seed0/sparse_deletions/default/100", "C3/seed0/dense_deletions/default/100",
                                  "sweep/seed5/sparse_deletions/L=4/100", "sweep/seed5/sparse_deletions/L=6/100",
                                  "sweep/seed5/sparse_deletions/L=8/cond_mode=E-marginal_term=True/100", "sweep/seed5/dense_deletions/L=8/50"])
def test_batched_recovery_of_windows_with_deletions_at_full_depth(name, monkeypatch):
    """Windows with five-candidate positions as a batch of one (gh_batch_spin): whatever flow the batch gives them -- the
    pipeline where it carries them, the batched launches over the five-symbol table where it does not -- against the digest."""
    from gretel_amd.hansel import Hansel, HanselBatch
    monkeypatch.setenv("GH_PIPE_MIN", "1")
    want = DIGESTS[name]
    case = want["case"]
    t = _table(case)
    h = Hansel(t.n_snps, band=t.band, **case["spec"])
    h.fill_from_support(t.rank, t.off, t.bases)
    if case.get("L"):
        h.L = case["L"]
    b = HanselBatch([h])
    r = b.spin(case["paths"])[0]
    _compare(mk.digest_result(r, h.export_band()), want, name + " as a batch")


def test_the_digest_file_covers_what_it_says():
    names = set(DIGESTS)
    assert "C5/seed0/default/1000" in names and DIGESTS["C5/seed0/default/1000"]["n"] == 1000
    assert all("C3/seed%d/default/100" % s in names for s in range(8))
    assert sum(1 for k in names if k.startswith("C3/seed0/cond_mode=")) == 19      # + the default spec = spec_matrix's 20
    assert "C3/seed0/sparse_deletions/default/100" in names
    # round 6: deletions beyond five lags, dense deletions, states as bytes
    assert all("sweep/seed5/sparse_deletions/L=%d/100" % L in names for L in (3, 4, 6, 7, 8, 11, 16, 22))
    assert all("sweep/seed5/dense_deletions/L=%d/50" % L in names for L in (4, 6, 8, 12))
    assert all("sweep/seed5/none/L=%d/50" % L in names for L in (33, 40, 48))
    assert {"C3/seed0/dense_deletions/default/100", "C3/seed0/dense_deletions/cond_mode=E-marginal_term=True/100"} <= names
    assert {c[0] for c in mk.cases()} == names
