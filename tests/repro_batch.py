"""Test infrastructure (calls the oracle; run from the repo root on the GPU box).  Batched recovery under candidate orders: which route / switch mismatches the C oracle (GPU box, repo root)."""
import os, sys, itertools
sys.path.insert(0, ".")
import numpy as np
from gretel_amd.hansel import Hansel, HanselBatch
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle
bad = {}
for route, order, mode, storage, L, mt in itertools.product(("streams", "batched"), ("ACGT-", "TG-AC", "-TGCA"), "AB", ("f32", "f64"), (1, 3), (False, True)):
    if route == "batched":
        os.environ["GH_BATCH_STREAMS_MAX"] = "-1"
    else:
        os.environ.pop("GH_BATCH_STREAMS_MAX", None)
    sw = dict(cand_order=order)
    ok = True
    for seed in range(4):
        ts = [make_support_table(17, 340, k=7, n_haps=8, err=0.05, seed=seed * 10 + q) for q in range(3)]
        hs, os_ = [], []
        for x in ts:
            hh = Hansel(x.n_snps, band=x.band, storage=storage, cond_mode=mode, marginal_term=mt, **sw)
            oo = COracle(x.n_snps, x.band, storage, mode, mt, **sw)
            assert hh.fill_from_support(x.rank, x.off, x.bases) == oo.fill(x)
            hh.L = L; oo.L = L
            hs.append(hh); os_.append(oo)
        for w, (res, oo) in enumerate(zip(HanselBatch(hs).spin(2), os_)):
            ref = oo.spin(2)
            if not (res["n"] == ref["n"] and np.array_equal(res["paths"], ref["paths"]) and res["hp_current"].tolist() == ref["hp_current"].tolist()):
                ok = False
                key = (route, order, mode, storage, L, mt)
                if key not in bad:
                    bad[key] = (seed, w, [bytes(b"ACGTN-_"[q] for q in p).decode() for p in res["paths"]], [bytes(b"ACGTN-_"[q] for q in p).decode() for p in ref["paths"]],
                                res["hp_current"].tolist(), ref["hp_current"].tolist())
for k, v in bad.items():
    print(k, v)
print("bad combos:", len(bad))
