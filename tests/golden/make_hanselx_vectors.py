#!/usr/bin/env python3
"""
PINNING KIT for the Hansel arithmetic -- run this where the REAL `hansel` module (hanselx==0.0.92, reference
setup.py:8) can be imported, together with the reference's own gretel/gretel.py:

    pip install hanselx==0.0.92
    python tests/golden/make_hanselx_vectors.py --reference /path/to/SamStudio8/gretel

It writes tests/golden/hanselx_vectors.json: inputs (support tables, as data) and what the real package answers for
them -- storage dtype, get_counts_at (keys AND their order), get_marginal_of_at, get_edge_weights_at (values AND key
order), reweight_observation return values, and five paths of gretel.generate_path / reweight_hansel_from_path run the
way gretel/cmd.py:148-179 runs them.  tests/test_hanselx_pinning.py then tries every oracle.hansel_ref.HanselSpec
against these vectors and names the spec that reproduces them: the day this file exists, "parity unpinned" ends.

It cannot run in the build container (no network, hanselx absent: SURVEY.md section 8(c)), it never travels to the GPU
box, and nothing in the product imports it.  Only DATA is written: no source text of hanselx or of the reference.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SYMBOLS = ['A', 'C', 'G', 'T', 'N', '-', '_']      # gretel/util.py:83
UNSYMBOLS = ['N', '_']
MIN_REMOVE = 0.01                                  # gretel/cmd.py:157


def fx(v):
    """A number as JSON can keep it exactly: [repr, hex of the binary64 value]."""
    v = float(v)
    return [repr(v), v.hex()]


def windows():
    """(name, n_snps, reads): the reference's own fixture (tests/data/test.sam as its support table)
    and two small synthetic windows (gretel_amd.synth: NumPy only)."""
    from gretel_amd.synth import make_support_table
    out = [("fixture", 4, [(0, "AAA"), (0, "CCC"), (0, "TT"), (0, "TT"), (2, "GG")])]
    t = make_support_table(40, 900, k=4, seed=3)
    out.append(("synth_40_k4", t.n_snps, list(t.reads())))
    t = make_support_table(70, 1000, k=None, seed=7, k_max=8)
    bases = t.bases.copy()
    import numpy as np
    bases[np.random.default_rng(5).random(len(bases)) < 0.06] = ord('-')      # some positions with five candidates
    t.bases = bases
    out.append(("synth_70_kvar_dels", t.n_snps, list(t.reads())))
    # reads of 7 SNPs: util.py:333 sets L = 7 -- beyond the lag counts the HIP path enumerates (its candidate pools walk this one)
    t = make_support_table(60, 700, k=7, seed=11)
    out.append(("synth_60_k7_L7", t.n_snps, list(t.reads())))
    # reads of up to 13 SNPs: observations up to 12 positions apart (a band wider than the 8-lane reweight groups), L = 10
    t = make_support_table(64, 420, k=None, seed=13, k_lambda=10.0, k_min=6, k_max=13)
    out.append(("synth_64_k13_band12", t.n_snps, list(t.reads())))
    return out


def fill(hansel, reads, n_snps):
    """gretel/util.py:226-286, 329-333 restated over a support table (oracle/gretel_ref.py: only add_observation and the
    three attributes are touched, so it drives the real Hansel exactly as load_from_bam does)."""
    from oracle.gretel_ref import fill_from_support
    return fill_from_support(hansel, reads, n_snps)


def ordered(d):
    return [[str(k), fx(v)] for k, v in d.items()]


def _dense(h):
    import numpy as np
    return h.dense() if hasattr(h, "dense") else np.asarray(h)


def dump_window(hansel_cls, generate_path, reweight_hansel_from_path, name, n, reads, n_paths, **init_kw):
    """Everything the pinning test compares, for one window, from ANY object with the hansel.Hansel surface
    (the real package here; the oracle in tests/test_hanselx_pinning.py's self-check)."""
    import numpy as np
    h = hansel_cls.init_matrix(SYMBOLS, UNSYMBOLS, n, **init_kw)
    stats = fill(h, reads, n)
    dense = _dense(h)
    w = dict(name=name, n_snps=n, reads=[[r, s] for r, s in reads], stats=[int(x) for x in stats], L=int(h.L),
             dtype=str(dense.dtype), shape=list(dense.shape))
    sym = h.symbols_d['A']
    w["symbol_type"] = type(sym).__name__
    w["symbol_equals_str"] = bool(sym == 'A')
    w["symbol_hash_equals_str"] = bool(hash(sym) == hash('A'))
    probe = sorted(set([0, 1, 2, 3, n // 2, n - 1, n]))
    w["counts_at"] = {str(p): ordered(h.get_counts_at(p)) for p in probe}
    w["marginal_of_at"] = {str(p): {s: fx(h.get_marginal_of_at(s, p)) for s in "ACGT-"} for p in probe}
    # edge weights: under the sentinel, under short and full histories (symbols as objects, like gretel.py:155)
    hist = [h.symbols_d['_']] + [h.symbols_d[c] for c in ("ACGT" * (n // 4 + 1))[:n]]
    w["edge_weights_at"] = {str(p): ordered(h.get_edge_weights_at(p, hist[:p])) for p in probe if p >= 1}
    # reweight_observation on a copy: return value and the cell afterwards
    hc = h.copy()
    rw = []
    for (a, b, i, j, ratio) in [('A', 'A', 1, 2, 0.25), ('C', 'C', 1, 2, 0.01), ('A', 'C', 1, 2, 0.5), ('_', 'A', 0, 1, 0.3)]:
        before = hc.get_observation(a, b, i, j)
        ret = hc.reweight_observation(a, b, i, j, ratio)
        rw.append(dict(cell=[a, b, i, j], ratio=fx(ratio), before=fx(before), returned=fx(ret), after=fx(hc.get_observation(a, b, i, j))))
    w["reweight_observation"] = rw
    # the spin loop of gretel/cmd.py:148-179
    orig = h.copy()
    recs = []
    for _ in range(n_paths):
        path, prob, mn = generate_path(n, h, orig)
        if path is None:
            break
        ratio = mn if mn >= MIN_REMOVE else MIN_REMOVE
        mag = reweight_hansel_from_path(h, path, ratio)
        recs.append(dict(path="".join(str(x) for x in path), hp_current=fx(prob["hp_current"]), hp_original=fx(prob["hp_original"]),
                         min_marginal=fx(mn), ratio=fx(ratio), magnitude=fx(mag)))
    w["records"] = recs
    w["final_sum"] = fx(_dense(h).astype("float64").sum())
    w["counts_at_after"] = {str(p): ordered(h.get_counts_at(p)) for p in probe}
    return w


def import_reference_gretel(reference):
    """gretel/gretel.py needs `hansel` and numpy; its package also imports pysam and PyVCF (gretel/util.py:1,5), which the
    two functions used here never touch: where those are missing, empty stand-in modules let the import through (the
    reference's own docs build does the same, docs/conf.py:24-27) -- recorded in the output."""
    import types
    stubbed = []
    sys.path.insert(0, reference)
    for mod in ("pysam", "vcf"):
        try:
            __import__(mod)
        except ImportError:
            sys.modules[mod] = types.ModuleType(mod)
            stubbed.append(mod)
    from gretel import gretel as ref_gretel
    return ref_gretel, stubbed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of SamStudio8/gretel (for gretel/gretel.py)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "hanselx_vectors.json"))
    ap.add_argument("--paths", type=int, default=5)
    args = ap.parse_args()

    import numpy as np
    from hansel import Hansel                      # the REAL one
    import hansel as hansel_mod
    ref_gretel, stubbed = import_reference_gretel(args.reference)

    vec = dict(note="produced by the real hanselx + the reference's gretel/gretel.py; see make_hanselx_vectors.py",
               hansel_version=getattr(hansel_mod, "__version__", "unknown"), numpy_version=np.__version__,
               stubbed_modules=stubbed, windows=[])
    for name, n, reads in windows():
        vec["windows"].append(dump_window(Hansel, ref_gretel.generate_path, ref_gretel.reweight_hansel_from_path, name, n, reads, args.paths))
    with open(args.out, "w") as fh:
        json.dump(vec, fh, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
