"""DFM06 / DFM09 / DFM17 telemetry text / JSON (the reference's conf_out / dat_out / print_gpx, dfm09mod.c:347-1150) —
include/sonde_dfm.h, host side.

Packet streams at the symbol level (`dfm09mod --softin`, no samples, no GPU): a DFM09, an inverted DFM17 (serial >= 23000000:
polarity-dependent type guess), a DFM06 (6-nibble serial in channel 6), a pressure-type sonde with bit errors (corrected,
uncorrectable, bursts) and positioning mode 3 — each through eight option sets (-v, -vv, --ecc, --ecc2, --ptu, --dist, --json,
--jsn_cfq, --sat, --auto, -i, -r --json).  Golden = stdout of the compiled reference on the same symbols
(tools/make_golden.py gen_dfm_fields); byte for byte."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402


@pytest.mark.parametrize("name", sorted(make_golden.DFM_FIELD_SCENARIOS))
def test_cli_dfm_telemetry_matches_reference(name):
    from radiosonde_auto_rx_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    gold = np.load(os.path.join(ROOT, "tests", "golden", "dfm_fields.npz"))
    soft = make_golden.dfm_field_symbols(make_golden.DFM_FIELD_SCENARIOS[name]).tobytes()
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    total = 0
    for k, args in enumerate(make_golden.DFM_FIELD_ARGS):
        r = subprocess.run([os.path.join(ROOT, "host", "bin", "dfm09mod")] + args + ["--softin"], input=soft, capture_output=True, env=env, timeout=120)
        want = gold["%s|%d" % (name, k)].tobytes()
        assert r.returncode == 0
        assert r.stdout == want, (name, args)
        total += len(want)
    assert total > 5000


NEW_ARGS = [["-vvv", "--ptu", "--dbg", "--ecc", "--auto"], ["-vvv", "--ecc", "--ptu", "-i"], ["--dbg", "--ptu", "-v"], ["--rawecc", "--auto"], ["--rawecc", "--json", "--auto", "--ecc"],
            ["-R", "--ecc", "--auto"], ["-R", "-i"], ["-vvv", "--dbg", "--ptu", "--ecc2", "--sat", "--auto"]]


@pytest.mark.parametrize("name", sorted(make_golden.DFM_FIELD_SCENARIOS))
def test_cli_dfm_verbose3_dbg_rawecc_match_compiled_reference(name):
    """-vvv (sensor type / polarity, battery, internal temperature, on-time), --dbg (measurement channels, the two alternative thermistor
    evaluations, resistor estimates), --rawecc and -R on the same scenarios: stdout of the compiled reference, byte for byte"""
    ref = os.path.join(ROOT, "oracle", "_ref", "dfm09mod")
    if not os.path.exists(ref):
        pytest.skip("compiled reference not present")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    soft = make_golden.dfm_field_symbols(make_golden.DFM_FIELD_SCENARIOS[name]).tobytes()
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    total = 0
    for args in NEW_ARGS:
        a = subprocess.run([os.path.join(ROOT, "host", "bin", "dfm09mod")] + args + ["--softin"], input=soft, capture_output=True, env=env, timeout=120)
        b = subprocess.run([ref] + args + ["--softin"], input=soft, capture_output=True, timeout=120)
        assert a.returncode == b.returncode == 0
        if a.stdout != b.stdout:
            for x, y in zip(a.stdout.splitlines(), b.stdout.splitlines()):
                assert x == y, (name, args, x, y)
        assert a.stdout == b.stdout, (name, args)
        total += len(a.stdout)
    assert total > 3000
