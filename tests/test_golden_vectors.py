"""Committed golden vectors (tests/golden/oracle_vectors.json, made by make_golden.py with the
Python oracle).  CPU: both oracles still reproduce them.  GPU: the HIP path reproduces them."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import gretel_ref as G
from oracle.c_oracle import COracle, paths_to_str
from oracle.hansel_ref import Hansel, HanselSpec, SYMBOLS, UNSYMBOLS

VEC = json.load(open(os.path.join(GOLDEN, "oracle_vectors.json")))["cases"]


class _T:
    def __init__(self, reads):
        self.rank = np.array([r for r, _ in reads], dtype=np.int32)
        self.off = np.concatenate([[0], np.cumsum([len(s) for _, s in reads])]).astype(np.int64)
        self.bases = np.frombuffer("".join(s for _, s in reads).encode(), dtype=np.uint8).copy()
        self.band = max(1, max(len(s) for _, s in reads) - 1)


@pytest.mark.parametrize("case", VEC, ids=[c["name"] for c in VEC])
def test_python_oracle_reproduces_golden(case):
    spec = HanselSpec(**case["spec"])
    h = Hansel.init_matrix(SYMBOLS, UNSYMBOLS, case["n_snps"], spec)
    assert list(G.fill_from_support(h, [tuple(r) for r in case["reads"]], case["n_snps"])) == case["stats"]
    assert h.L == case["L"]
    for p, c in case["counts"].items():
        assert [float(x) for x in h._counts(int(p))] == c
    recs, _ = G.recover_paths(h, case["n_snps"], len(case["records"]))
    assert recs == case["records"]
    assert float(h.dense().astype("float64").sum()) == case["final_sum"]


@pytest.mark.parametrize("case", VEC, ids=[c["name"] for c in VEC])
def test_c_oracle_reproduces_golden(case):
    t = _T([tuple(r) for r in case["reads"]])
    sp = case["spec"]
    o = COracle(case["n_snps"], t.band, use_libm=True, **sp)
    assert list(o.fill(t)) == case["stats"]
    r = o.spin(len(case["records"]))
    assert paths_to_str(r["paths"]) == [x["path"] for x in case["records"]]
    assert r["hp_current"].tolist() == [x["hp_current"] for x in case["records"]]
    assert r["magnitude"].tolist() == [x["magnitude"] for x in case["records"]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", VEC, ids=[c["name"] for c in VEC])
def test_hip_reproduces_golden(case):
    from gretel_amd.hansel import Hansel as DevHansel
    t = _T([tuple(r) for r in case["reads"]])
    sp = case["spec"]
    h = DevHansel(case["n_snps"], band=t.band, **sp)
    assert list(h.fill_from_support(t.rank, t.off, t.bases)) == case["stats"]
    assert h.L == case["L"]
    for p, c in case["counts"].items():
        assert h.counts_array(int(p))[:7].tolist() == c
    res = h.spin(len(case["records"]))
    assert [DevHansel.path_str(p) for p in res["paths"]] == [x["path"] for x in case["records"]]
    # libm's log10 made the vectors, the kernels' log10 (include/gh_detlog.h) is that function: the very doubles (bar: 1e-6)
    assert np.array_equal(res["hp_current"], np.array([x["hp_current"] for x in case["records"]], dtype=float), equal_nan=True)
    assert np.array_equal(res["hp_original"], np.array([x["hp_original"] for x in case["records"]], dtype=float), equal_nan=True)
    assert res["ratio"].tolist() == [x["ratio"] for x in case["records"]]
    assert np.allclose(res["magnitude"], [x["magnitude"] for x in case["records"]], rtol=1e-12, atol=0)
    assert abs(h.export_band().sum() - case["final_sum"]) <= 1e-9 * case["final_sum"]


HANSELX = os.path.join(GOLDEN, "hanselx_vectors.json")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(HANSELX), reason="tests/golden/hanselx_vectors.json absent (made by make_hanselx_vectors.py where "
                    "hanselx==0.0.92 is importable): the HIP path is held to the oracle only until then")
def test_hip_reproduces_the_real_hanselx_vectors():
    """The day the pinning kit has run: what the REAL package and the reference's own gretel.py answered (five paths per
    window: the fixture, L = 4, five-candidate positions, L = 7 -- the candidate pools --, a band of 12), through the product path
    under its default spec.  Paths bit-exact, likelihoods within 1e-6 (BASELINE.json), ratios within 1e-12."""
    from gretel_amd.hansel import Hansel as DevHansel
    vec = json.load(open(HANSELX))
    for w in vec["windows"]:
        t = _T([(int(r), s) for r, s in w["reads"]])
        h = DevHansel(w["n_snps"], band=t.band)
        assert list(h.fill_from_support(t.rank, t.off, t.bases)) == w["stats"], w["name"]
        assert h.L == w["L"], w["name"]
        res = h.spin(len(w["records"]) or 1)
        assert [DevHansel.path_str(p) for p in res["paths"]] == [x["path"] for x in w["records"]], w["name"]
        want = lambda k: np.array([float.fromhex(x[k][1]) for x in w["records"]])
        assert np.allclose(res["hp_current"], want("hp_current"), rtol=0, atol=1e-6), w["name"]
        assert np.allclose(res["hp_original"], want("hp_original"), rtol=0, atol=1e-6), w["name"]
        assert np.allclose(res["ratio"], want("ratio"), rtol=1e-12, atol=0), w["name"]
        assert np.allclose(res["magnitude"], want("magnitude"), rtol=1e-9, atol=0), w["name"]
