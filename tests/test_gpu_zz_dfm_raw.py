"""`dfm09mod --rawecc` (the frame bits before the Hamming decoder — what auto_rx asks for when it saves raw frames, decode.py:1078), `-R` (packet hex) and
`-vvv --dbg` on SAMPLES against the compiled reference.  Frame 0 of a header hit takes its 16 header bits from the previous frame, as the reference's buffer
does.  (The soft-bit forms of the same options are compared on the CPU: tests/test_softin.py, tests/test_dfm_fields.py.)"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dfm_raw_forms_on_samples_match_reference():
    from golden_cases import DFM_NAMES, dfm_capture
    ref = os.path.join(ROOT, "oracle", "_ref", "dfm09mod")
    if not os.path.exists(ref):
        pytest.skip("compiled reference not present")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    x, fq, sr, ecc = dfm_capture(DFM_NAMES[0])
    tail = ["--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"]
    for args in (["--rawecc", "--auto"], ["-R", "--ecc", "--auto"], ["-vvv", "--ecc", "--ptu", "--dbg", "--auto"]):
        a = subprocess.run([os.path.join(ROOT, "host", "bin", "dfm09mod")] + args + tail, input=x.tobytes(), capture_output=True, timeout=120)
        b = subprocess.run([ref] + args + tail, input=x.tobytes(), capture_output=True, timeout=120)
        assert a.returncode == b.returncode == 0 and a.stdout == b.stdout, (args, a.stdout[:300], b.stdout[:300])
