"""RS92 text / JSON incl. the GPS solution against golden stdout of the compiled reference (tests/golden/rs92_fields.npz, made by
tools/make_golden.py gen_rs92_fields): the native `rs92mod --softin` on the same soft-symbol streams with the same orbit files, both re-created
from seeds.  Runs where the compiled reference is not present (tests/test_rs92_native.py compares with it directly where it is).  No GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402


@pytest.fixture(scope="module")
def orbits(tmp_path_factory):
    from radiosonde_auto_rx_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    return make_golden.rs92_orbit_files(str(tmp_path_factory.mktemp("rs92gold")))


@pytest.mark.parametrize("name", sorted(make_golden.RS92_FIELD_SCENARIOS))
def test_cli_rs92_text_and_positions_match_the_golden_reference_output(orbits, name):
    eph, E, A = orbits
    gold = np.load(os.path.join(ROOT, "tests", "golden", "rs92_fields.npz"))
    soft = make_golden.rs92_field_symbols(make_golden.RS92_FIELD_SCENARIOS[name], eph).tobytes()
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    total = 0
    for k, args in enumerate(make_golden.RS92_FIELD_ARGS):
        r = subprocess.run([os.path.join(ROOT, "host", "bin", "rs92mod")] + make_golden.rs92_field_args(args, E, A) + ["--softin"], input=soft, capture_output=True,
                           env=env, timeout=120)
        want = gold["%s|%d" % (name, k)].tobytes()
        assert r.returncode == 0
        assert r.stdout == want, (name, args)
        total += len(want)
    assert total > 10000
