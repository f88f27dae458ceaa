"""M10 / M10+ telemetry text / JSON (the reference's m10mod print_pos(), m10mod.c:862-1047) — include/sonde_m10.h — and the
soft-symbol framer behind `m10mod --softin` (header threshold 0.8, two symbols per bit, differential decoding, rest of the second
dropped unless -vvv), host side, no GPU.

Frame streams at the symbol level: Trimble-GPS frames (week rollover repair, three thermistor ranges), Gtop-GPS frames, frames
with a bad checksum or an implausible week — each through seven option sets (-v / -vv / -vvv, --ptu, --json, --jsn_cfq, -r).
Golden = stdout of the compiled reference on the same symbols (tools/make_golden.py gen_m10_fields); byte for byte, including the
-vvv quirk of the reference's on-chip temperature (see sonde_m10_fields.cpp)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402


@pytest.mark.parametrize("name", sorted(make_golden.M10_FIELD_SCENARIOS))
def test_cli_m10_telemetry_matches_reference(name):
    from radiosonde_auto_rx_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    gold = np.load(os.path.join(ROOT, "tests", "golden", "m10_fields.npz"))
    soft = make_golden.m10_field_symbols(make_golden.M10_FIELD_SCENARIOS[name]).tobytes()
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    total = 0
    for k, args in enumerate(make_golden.M10_FIELD_ARGS):
        r = subprocess.run([os.path.join(ROOT, "host", "bin", "m10mod")] + args + ["--softin"], input=soft, capture_output=True, env=env, timeout=120)
        want = gold["%s|%d" % (name, k)].tobytes()
        assert r.returncode == 0
        assert r.stdout == want, (name, args)
        total += len(want)
    assert total > 5000
