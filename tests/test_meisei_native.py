"""Meisei iMS-100 / RS-11G bit-rate tier (include/sonde_meisei.h, host/meisei100mod.c --softin): stdout byte for byte against the compiled
reference (`oracle/_ref/meisei100mod`) on the same float32 soft half-symbol streams (decode.py:1379: `meisei100mod --softin --json --ptu --ecc`).
The frames come from tools/synth.py (BCH(63,51) blocks, parity bits, GPS checksum, 64-word configuration cycle); the reference accepting
them — (ok)[OK] on every line of a clean stream — is what pins the generator.  No GPU involved."""
import json
import os
import subprocess

import numpy as np
import pytest

from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "meisei100mod")
BIN = os.path.join(ROOT, "host", "bin", "meisei100mod")

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present (oracle/Makefile builds it where /root/reference exists)")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "radiosonde_auto_rx_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])


def _soft(sym, sigma=0.0, seed=1, lead=29, invert=False, cut=None):
    rng = np.random.default_rng(seed)
    s = np.concatenate([rng.normal(0, 0.3, lead), 2.0 * sym.astype(np.float64) - 1.0])
    s = s + rng.normal(0.0, sigma, len(s))
    if invert:
        s = -s
    if cut is not None:
        s = s[:cut]
    return s.astype(np.float32).tobytes()


def _both(args, data):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([BIN] + args, input=data, capture_output=True, timeout=120, env=env)
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=120)
    assert a.returncode == b.returncode, (args, a.stderr[-300:], b.stderr[-300:])
    if a.stdout != b.stdout:
        for x, y in zip(a.stdout.decode().splitlines(), b.stdout.decode().splitlines()):
            assert x == y, (args, x, y)
    assert a.stdout == b.stdout
    return a.stdout.decode()


OPTS = [["--softin"], ["--softin", "--ecc"], ["--softin", "--ecc", "-v", "--ptu"], ["--softin", "--json", "--ptu", "--ecc"], ["--softin", "-r"], ["--softin", "-r", "--ecc", "-v"],
        ["--softin", "--dbg", "--ecc"], ["--softin", "--json", "--jsn_cfq", "404500000", "--year", "2031"], ["--softin", "--ims100", "--ecc", "--ptu"],
        ["--softin", "--rs11g", "--ecc", "--ptu", "-v"]]


@pytest.mark.parametrize("variant", ["ims100", "rs11g"])
@pytest.mark.parametrize("args", OPTS, ids=lambda a: " ".join(a))
def test_meisei_clean_and_noisy(variant, args):
    sym = synth.meisei_symbols(70, variant)
    out = _both(args, _soft(sym))
    if "--ecc" in args and "-r" not in args and variant == "ims100":
        assert out.count("(ok)[OK]") >= 68
    for sigma, seed in ((0.35, 2), (0.5, 3), (0.7, 4)):
        _both(args, _soft(sym, sigma=sigma, seed=seed))


@pytest.mark.parametrize("variant,sub,idp", [("ims100", "IMS100", "IMS100-4123456"), ("rs11g", "RS11G", "RS11G-4123456")])
def test_meisei_json_fields_and_ptu(variant, sub, idp):
    """a full configuration cycle (64 frames) gives serial number, transmit frequency and the sensor calibration: temp / humidity appear"""
    out = _both(["--softin", "--json", "--ptu", "--ecc"], _soft(synth.meisei_symbols(140, variant), sigma=0.2))
    js = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(js) >= 60
    d = js[-1]
    assert d["type"] == "MEISEI" and d["subtype"] == sub and d["id"] == idp and d["version"] == "oracle"
    assert "temp" in d and -90 < d["temp"] < 40 and 0 < d["humidity"] < 100 and d["tx_frequency"] in (404250, 405000)
    assert abs(d["lat"] - 35.2) < 0.1 and abs(d["lon"] - 139.6) < 0.3 and d["datetime"].startswith("2024-06-15T12:0")


def test_variant_hand_over_both_ways():
    """the decoder starts as RS-11G on stdin (the default applies only with a file name, meisei100mod.c:551-553), sees the iMS-100 type word and
    switches; an RS-11G sonde after that switches it back; state is reset at each change"""
    data = _soft(synth.meisei_symbols(20, "ims100"), sigma=0.2) + _soft(synth.meisei_symbols(20, "rs11g", k0=40), sigma=0.2, seed=5, lead=0) \
        + _soft(synth.meisei_symbols(8, "ims100", k0=90), sigma=0.2, seed=6, lead=0)
    for args in (["--softin", "--ecc", "-v", "--ptu"], ["--softin", "--json"], ["--softin", "--ims100", "--ecc"], ["--softin"]):
        out = _both(args, data)
    assert "lat: 35.2" in out and "lat: 35.1" in out


def test_inverted_and_softinv():
    sym = synth.meisei_symbols(12, "ims100")
    a = _both(["--softin", "--ecc"], _soft(sym, sigma=0.2, invert=True))
    b = _both(["--softinv", "--ecc"], _soft(sym, sigma=0.2, invert=True))
    assert a.count("[OK]") >= 10 and b.count("[OK]") >= 10


@pytest.mark.parametrize("cut", [10, 29 + 48 + 5, 29 + 48 + 1151, 29 + 48 + 1152, 29 + 1200 * 3 - 1])
def test_truncated_streams(cut):
    sym = synth.meisei_symbols(6, "ims100")
    _both(["--softin", "--ecc"], _soft(sym, sigma=0.3, cut=cut))
    _both(["--softin", "-r"], _soft(sym, sigma=0.3, cut=cut))


def test_noise_only_and_empty():
    rng = np.random.default_rng(11)
    _both(["--softin", "--ecc"], rng.normal(0, 1, 30000).astype(np.float32).tobytes())
    assert _both(["--softin", "--ecc"], b"") == "\n"


def test_file_argument_sets_default_variant_and_ends_the_argument_list(tmp_path):
    p = tmp_path / "soft.f32"
    p.write_bytes(_soft(synth.meisei_symbols(6, "rs11g"), sigma=0.1))
    # with a file name the decoder starts as iMS-100 and what follows the name is ignored (here: --json)
    for args in (["--softin", "--ecc", str(p), "--json"], ["--softin", "-r", str(p)], ["--softin", "--rs11g", "--ecc", str(p)]):
        env = dict(os.environ, SONDE_JSN_VERSION="oracle")
        a = subprocess.run([BIN] + args, capture_output=True, timeout=60, env=env)
        b = subprocess.run([REF] + args, capture_output=True, timeout=60)
        assert a.returncode == b.returncode == 0 and a.stdout == b.stdout and b"{" not in a.stdout


def test_bad_options():
    assert subprocess.run([BIN, "--nonsense"], capture_output=True).returncode == 255
    assert subprocess.run([BIN, "--br"], capture_output=True).returncode == 255
    r = subprocess.run([BIN, "-", "48000", "16"], input=b"", capture_output=True)
    assert r.returncode == 255 and b"raw data not IQ" in r.stderr
    assert subprocess.run([BIN, "--help"], capture_output=True).returncode == 0
