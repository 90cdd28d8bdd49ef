"""tests/golden/fullsize_digests.json is what the C oracle gives TODAY (CPU, no GPU): two of its cases are regenerated and
compared, so a change of the oracle or of the seeded generator cannot leave stale digests behind unnoticed."""
import json
import os
import sys

import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import make_fullsize_digests as mk  # noqa: E402


@pytest.mark.parametrize("name", ["C3/seed0/default/100", "C3/seed0/cond_mode=E-marginal_term=True-storage=f64/100",
                                  "C3/seed0/sparse_deletions/default/100"])
def test_a_case_regenerates_to_the_committed_digest(name):
    with open(os.path.join(GOLDEN, "fullsize_digests.json")) as f:
        doc = json.load(f)
    want = doc["cases"][name]
    case = dict(mk.cases())[name]
    assert want["case"] == case
    _, got = mk.run_case((name, case))
    for key in ("n", "hole_at", "path_sha", "hp_current", "hp_original", "ratio", "magnitude", "band_sha", "fill_stats", "L"):
        assert got[key] == want[key], key
    assert "parity unpinned" in doc["meta"]["parity"]
