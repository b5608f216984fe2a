"""A bounded slice of the differential fuzzers (tests/fuzz/) inside `-m gpu`: in round 4 the
fuzzers found three bugs no parity test saw -- one a GPU memory fault reachable with a legal PatchSize -- but only ran in
hand-started campaigns.  Each test runs one fuzzer's own case generator and checks (random shapes, thresholds, arc lengths,
PatchSize / NumBRIEFPairs / suppression windows / strided views for the detector; set sizes, widths, live bits, duplicate
clusters, keypoint layouts and both scans for the matcher; mixed batches; pyramids; pose scenes incl. both geometry modes;
Gauss-Newton windows; the exact sort in its one- and two-launch forms) against the oracle with FIXED seeds and case counts sized for about a minute in total.  The long
campaigns stay in tests/fuzz/campaign.sh."""
import importlib
import os
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))


@pytest.mark.parametrize("name,cases,seed", [
    ("fuzz_detector", 300, 9001),
    ("fuzz_matcher", 300, 9002),
    ("fuzz_batch", 60, 9003),
    ("fuzz_pyramid", 40, 9004),
    ("fuzz_pose", 30, 9005),
    ("fuzz_gn", 20, 9006),
    ("fuzz_sort", 150, 9007),
])
def test_fuzz_smoke(name, cases, seed, capsys):
    mod = importlib.import_module(name)
    t0 = time.perf_counter()
    mod.main(cases, seed)  # raises (AssertionError with the case's parameters) at the first difference
    dt = time.perf_counter() - t0
    out = capsys.readouterr().out
    assert "cases" in out, out[-300:]  # the fuzzer's own summary line was printed
    assert dt < 120.0, "%s: %d cases took %.0f s -- resize the smoke" % (name, cases, dt)
