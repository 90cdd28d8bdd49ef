import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_rw_lanes import _table, _pair
t = _table(21, 12.0, seed=515)
def cmp(tag, r, q):
    print(tag, "paths", np.array_equal(r["paths"], q["paths"]), "hpc", r["hp_current"].tolist() == q["hp_current"].tolist(), "hpo", r["hp_original"].tolist() == q["hp_original"].tolist(), r["hp_original"][:1], q["hp_original"][:1], flush=True)
for cond in ("A", "C"):
    h, o = _pair(t, 11, cond_mode=cond)
    h.snapshot_original(); o.snapshot_original()
    cmp(cond + " spin1", h.spin(6), o.spin(6))
    syms = "ACGT"
    for k, (a, b, i) in enumerate([(0, 1, 40), (2, 2, 41), (3, 0, 700), (1, 3, 701)]):
        for _ in range(3 + k):
            h.add_observation(syms[a], syms[b], i, i + 1 + k)
            o.add(a, b, i, i + 1 + k)
    cmp(cond + " after adds", h.spin(5), o.spin(5))
    print(h.reweight_observation("A", "C", 40, 41, 0.5), o.reweight_obs(0, 1, 40, 41, 0.5))
    cmp(cond + " after rw_obs", h.spin(5), o.spin(5))
    p, _ = o.generate_path()
    got = h.generate_path()
    print("gen", np.array_equal(got[0], p))
    print(h.reweight_from_path(got[0], 0.25), o.reweight_path(p, 0.25))
    cmp(cond + " after rw_path", h.spin(5), o.spin(5))
    print("band", np.array_equal(h.export_band(), o.export_band()))
