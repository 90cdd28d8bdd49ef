"""Helpers shared by the GPU parity tests: a device Hansel and the C oracle built from the same switches
(oracle.hansel_ref.HanselSpec's names), filled from the same support table."""
import contextlib
import os

import numpy as np

ORACLE_KEYS = ("storage", "cond_mode", "marginal_term", "cand_order", "offer_zero")


@contextlib.contextmanager
def walk_mode(mode):
    """GH_WALK is read when a handle is created."""
    old = os.environ.get("GH_WALK")
    if mode is None:
        os.environ.pop("GH_WALK", None)
    else:
        os.environ["GH_WALK"] = mode
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("GH_WALK", None)
        else:
            os.environ["GH_WALK"] = old


def make_pair(t, L=None, walk=None, **kw):
    from gretel_amd.hansel import Hansel
    from oracle.c_oracle import COracle
    with walk_mode(walk):
        h = Hansel(t.n_snps, band=t.band, **kw)
    o = COracle(t.n_snps, t.band, **{k: v for k, v in kw.items() if k in ORACLE_KEYS})
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    if L is not None:
        h.L = L
        o.L = L
    return h, o


def same(res, ref, rtol=1e-10):
    assert res["n"] == ref["n"] and res["hole_at"] == ref["hole_at"], (res["n"], ref["n"], res["hole_at"], ref["hole_at"])
    assert np.array_equal(res["paths"], ref["paths"]), "recovered SNP sequences differ (first at path %d)" % next(
        q for q in range(res["n"]) if not np.array_equal(res["paths"][q], ref["paths"][q]))
    assert res["hp_current"].tolist() == ref["hp_current"].tolist()
    assert res["hp_original"].tolist() == ref["hp_original"].tolist()
    assert res["ratio"].tolist() == ref["ratio"].tolist()
    assert np.allclose(res["magnitude"], ref["magnitude"], rtol=rtol, atol=0)


def with_dels(t, frac, seed):
    bases = t.bases.copy()
    bases[np.random.default_rng(seed).random(len(bases)) < frac] = ord('-')
    t.bases = bases
    return t


def spec_id(kw):
    return "-".join("%s=%s" % (k, v) for k, v in sorted(kw.items())) or "default"
