"""InterMet iMet-54 / iMet-50 bit-rate tier (include/sonde_imet54.h, host/imet54mod.c --softin / --rawhex): stdout byte for byte against the
compiled reference (`oracle/_ref/imet54mod`) on the same float32 soft-bit streams (decode.py:1250: `imet54mod --ecc --json --softin -i --ptu`).
Frames from tools/synth.py (8N1, 64-bit interleave, Hamming(8,4), both frame checks); the reference printing [OK] / [ok] for them pins the
generator.  No GPU involved."""
import json
import os
import subprocess

import numpy as np
import pytest

from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "imet54mod")
BIN = os.path.join(ROOT, "host", "bin", "imet54mod")

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present (oracle/Makefile builds it where /root/reference exists)")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "radiosonde_auto_rx_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])


def _soft(bits, sigma=0.0, seed=1, invert=False, cut=None):
    rng = np.random.default_rng(seed)
    s = 2.0 * bits.astype(np.float64) - 1.0 + rng.normal(0.0, sigma, len(bits))
    if invert:
        s = -s
    if cut is not None:
        s = s[:cut]
    return s.astype(np.float32).tobytes()


def _both(args, data=None):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([BIN] + args, input=data, capture_output=True, timeout=120, env=env)
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=120)
    assert a.returncode == b.returncode, (args, a.stderr[-300:], b.stderr[-300:])
    assert a.stdout == b.stdout, (args, a.stdout[:600], b.stdout[:600])
    return a.stdout


OPTS = [["--softin"], ["--softin", "--ecc"], ["--softin", "--ecc", "-v", "--ptu"], ["--ecc", "--json", "--softin", "--ptu"], ["--softin", "-r"], ["--softin", "-r", "--ecc"],
        ["--softin", "-r4", "--ecc"], ["--softin", "-r", "--json", "--ptu"], ["--softin", "--json", "--jsn_cfq", "402300000", "--silent"]]


@pytest.mark.parametrize("check", ["std", "cont", "none"])
@pytest.mark.parametrize("args", OPTS, ids=lambda a: " ".join(a))
def test_imet54_clean_and_noisy(args, check):
    bits = synth.imet54_onair_bits(5, check=check)
    out = _both(args, _soft(bits))
    if "--silent" not in args and not ("-r" in args and "--json" in args):
        assert out.count({"std": b"[OK]", "cont": b"[ok]", "none": b"[oo]" if "--ecc" in args or "--json" in args else b"["}[check]) >= 5
    for sigma, seed in ((0.3, 2), (0.45, 3), (0.6, 4)):
        _both(args, _soft(bits, sigma=sigma, seed=seed))


def test_imet54_json_fields_and_imet50():
    out = _both(["--ecc", "--json", "--softin", "--ptu"], _soft(synth.imet54_onair_bits(4), sigma=0.1)).decode()
    js = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(js) == 4
    d = js[1]
    assert d["type"] == "IMET5" and d["id"] == "IMET5-54012345" and d["subtype"] == "iMet-54" and d["datetime"] == "12:34:01.250Z" and d["frame"] == 45241
    assert abs(d["lat"] - 52.12356) < 2e-5 and abs(d["alt"] - 2350.6) < 1e-6 and abs(d["temp"] + 12.6) < 1e-6 and 60 < d["humidity"] < 75 and d["version"] == "oracle"
    out = _both(["--ecc", "--json", "--softin", "--ptu"], _soft(synth.imet54_onair_bits(3, imet50=True), sigma=0.1)).decode()
    js = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(js) == 3 and js[0]["subtype"] == "iMet-50" and "temp" not in js[0]


def test_polarity_invert_and_auto():
    """a stream of the other polarity is skipped without --auto and decoded with it; -i expects it (imet54mod.c:1018-1021)"""
    bits = synth.imet54_onair_bits(4)
    assert _both(["--softin", "--ecc"], _soft(bits, sigma=0.2, invert=True)) == b""
    assert _both(["--softin", "--ecc", "--auto"], _soft(bits, sigma=0.2, invert=True)).count(b"[OK]") == 4
    assert _both(["--softin", "--ecc", "-i"], _soft(bits, sigma=0.2, invert=True)).count(b"[OK]") == 4
    assert _both(["--softinv", "--ecc"], _soft(bits, sigma=0.2, invert=True)).count(b"[OK]") == 4
    assert _both(["--softin", "--ecc", "-i"], _soft(bits, sigma=0.2)) == b""
    mixed = _soft(bits, sigma=0.2) + _soft(bits, sigma=0.2, seed=9, invert=True) + _soft(bits, sigma=0.2, seed=10)
    assert _both(["--softin", "--ecc", "--auto"], mixed).count(b"[OK]") == 12
    assert _both(["--softin", "--ecc"], mixed).count(b"[OK]") == 8


def test_rawhex_round_trip():
    raw = _both(["--softin", "-r", "--ecc"], _soft(synth.imet54_onair_bits(4, check="cont"), sigma=0.3))
    for args in (["--rawhex", "--ptu", "-v"], ["--rawhex", "--json", "--ptu"], ["--rawhex", "-r"], ["--rawhex", "-r4", "--ecc"]):
        out = _both(args, raw)
    assert out.count(b"[ok]") == 4
    _both(["--rawhex"], b"0102\n\nzz11223344556677889900aabbccddeeff00112233445566778899\n" + raw[:150] + b"\n")


@pytest.mark.parametrize("cut", [100, 130 + 200, 130 + 640, 130 + 1300, 130 + 2199, 4798 + 130 + 2100])
def test_truncated_streams(cut):
    """a frame cut short by the end of the stream is still printed from the bits that exist (the rest of the byte buffer is the previous frame's)"""
    bits = synth.imet54_onair_bits(3)
    for args in (["--softin", "--ecc", "--ptu"], ["--softin", "-r"]):
        _both(args, _soft(bits, sigma=0.2, cut=cut))


def test_noise_only_and_empty():
    rng = np.random.default_rng(11)
    _both(["--softin", "--ecc"], rng.normal(0, 1, 40000).astype(np.float32).tobytes())
    assert _both(["--softin", "-v"], b"") == b""


def test_bad_options():
    assert subprocess.run([BIN, "--nonsense"], capture_output=True).returncode == 255
    assert subprocess.run([BIN, "--br"], capture_output=True).returncode == 255
    r = subprocess.run([BIN, "-", "48000", "16"], input=b"", capture_output=True)
    assert r.returncode == 255 and b"raw data not IQ" in r.stderr
