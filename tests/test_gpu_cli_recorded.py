"""Keeps tests/golden/cli_ours.npz honest: the stdout / stderr / exit codes of host/bin/dft_detect and host/bin/fsk_demod on auto_rx's own
pipelines (tools/caller_cases.py), recorded on an MI355X by tools/record_cli_outputs.py, are what tests/test_caller_contract.py feeds to
auto_rx's parsers in the container that holds the reference.  Here, on the GPU, the binaries must still print exactly that."""
import os
import re
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.environ.get("SONDE_CLI_RECORDED", os.path.join(ROOT, "tests", "golden", "cli_ours.npz"))


def test_binaries_still_print_what_was_recorded():
    sys.path.insert(0, ROOT)
    from tools import record_cli_outputs as rec
    if not os.path.exists(os.path.join(ROOT, "host", "bin", "fsk_demod")):
        pytest.fail("host/bin is not built (run __graft_entry__.build())")
    want, got = rec.load(FIX), rec.run_all()
    assert sorted(want) == sorted(got)
    for k in sorted(want):
        a, b = want[k], got[k]
        assert a == b, (k, a if k.endswith(".rc") else next(((i, x[:120], y[:120]) for i, (x, y) in enumerate(zip(a.split(b"\n"), b.split(b"\n"))) if x != y), len(a)))
    assert any(re.search(rb"RS41: 0\.\d+", want[k]) for k in want if k.endswith("detect_rs41.stdout"))
