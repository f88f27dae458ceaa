"""Meteo-Radiy MRZ (MP3-H1) bit-rate tier (include/sonde_mrz.h, host/mp3h1mod.c --softin / --rawhex): stdout byte for byte against the compiled
reference (`oracle/_ref/mp3h1mod`) on the same float32 soft half-symbol streams (decode.py:1293: `mp3h1mod --auto --json --softin --ptu`).
Frames from tools/synth.py (both position formats, 16-word configuration cycle, CRC); the reference printing [OK] for them pins the generator.
The reference's header threshold (0.82 of a 1001.. preamble) also fires on partial matches; those frames come out [NO] on both sides.
No GPU involved."""
import json
import os
import subprocess

import numpy as np
import pytest

from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "mp3h1mod")
BIN = os.path.join(ROOT, "host", "bin", "mp3h1mod")

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present (oracle/Makefile builds it where /root/reference exists)")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "radiosonde_auto_rx_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])


def _soft(sym, sigma=0.0, seed=1, lead=61, invert=False, cut=None):
    rng = np.random.default_rng(seed)
    s = np.concatenate([2.0 * rng.integers(0, 2, lead) - 1.0, 2.0 * sym.astype(np.float64) - 1.0])
    s = s + rng.normal(0.0, sigma, len(s))
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
    if a.stdout != b.stdout:
        for x, y in zip(a.stdout.splitlines(), b.stdout.splitlines()):
            assert x == y, (args, x, y)
    assert a.stdout == b.stdout
    return a.stdout


OPTS = [["--softin"], ["--softin", "-v"], ["--softin", "-vv", "--ptu", "--dbg"], ["--auto", "--json", "--softin", "--ptu"], ["--softin", "-r"], ["--softin", "-R"],
        ["--softin", "--uniq", "--json", "--jsn_cfq", "403000000"], ["--softin", "-c", "--ptu"], ["--softin", "--ofs", "8", "-r"]]


@pytest.mark.parametrize("latlon", [False, True], ids=["ecef", "latlon"])
@pytest.mark.parametrize("args", OPTS, ids=lambda a: " ".join(a))
def test_mrz_clean_and_noisy(args, latlon):
    sym = synth.mrz_symbols(20, latlon=latlon)
    out = _both(args, _soft(sym))
    if "-R" not in args:
        assert out.count(b"[OK]") >= (10 if "--uniq" in args else 20)
    for sigma, seed in ((0.3, 2), (0.5, 3), (0.7, 4)):
        _both(args, _soft(sym, sigma=sigma, seed=seed))


@pytest.mark.parametrize("latlon,ref", [(False, "GPS"), (True, "MSL")], ids=["ecef", "latlon"])
def test_mrz_json_fields(latlon, ref):
    """JSON appears once the date (sub-frame 15) and both serial numbers (12, 13) have been received"""
    out = _both(["--json", "--softin", "--ptu", "--uniq"], _soft(synth.mrz_symbols(36, latlon=latlon), sigma=0.1)).decode()
    js = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(js) >= 3
    d = js[-1]
    assert d["type"] == "MRZ" and d["id"] == "MRZ-21043-18765" and d["datetime"] == "2024-06-15T12:34:35Z" and d["ref_position"] == ref and d["version"] == "oracle"
    assert abs(d["lat"] - 55.75474) < 2e-5 and abs(d["lon"] - 37.61492) < 2e-5 and abs(d["alt"] - 3631.7) < 0.05 and abs(d["temp"] + 12.7) < 1e-6
    assert ("vel_v" in d) == (not latlon) and d["sats"] == (9 if latlon else 11)


def test_frame_type_change_in_mid_stream():
    """ECEF frames, then lat / lon frames, then ECEF again: the number of bits read per header follows the last frame whose CRC held"""
    data = _soft(synth.mrz_symbols(6), sigma=0.2) + _soft(synth.mrz_symbols(6, latlon=True), sigma=0.2, seed=5) + _soft(synth.mrz_symbols(6), sigma=0.2, seed=6)
    for args in (["--softin", "-vv", "--ptu"], ["--softin", "-r"], ["--softin", "--json"]):
        out = _both(args, data)
    assert out.count(b"[OK]") >= 12


def test_polarity_invert_and_auto():
    sym = synth.mrz_symbols(6)
    assert _both(["--softin"], _soft(sym, sigma=0.2, invert=True)).count(b"[OK]") == 0
    assert _both(["--softin", "--auto"], _soft(sym, sigma=0.2, invert=True)).count(b"[OK]") >= 6
    assert _both(["--softin", "-i"], _soft(sym, sigma=0.2, invert=True)).count(b"[OK]") >= 6
    assert _both(["--softinv"], _soft(sym, sigma=0.2, invert=True)).count(b"[OK]") >= 6
    mixed = _soft(sym, sigma=0.2) + _soft(sym, sigma=0.2, seed=9, invert=True) + _soft(sym, sigma=0.2, seed=10)
    assert _both(["--softin", "--auto"], mixed).count(b"[OK]") >= 18


def test_rawhex_round_trip():
    raw = _both(["--softin", "-r"], _soft(synth.mrz_symbols(18), sigma=0.3))
    for args in (["--rawhex", "--ptu", "-vv"], ["--rawhex", "--json", "--ptu"], ["--rawhex", "-r"]):
        out = _both(args, raw)
    assert out.count(b"[OK]") >= 18
    _both(["--rawhex"], b"01 02\n\nzz 11 22 33 44 55 66 77 88 99 00 aa bb cc dd ee ff 00 11 22 33 44 55 66 77 88 99\n" + raw[:100] + b"\n")


@pytest.mark.parametrize("cut", [30, 61 + 44 + 100, 61 + 44 + 2 * 200, 61 + 44 + 2 * 385 + 1, 61 + 2399 + 44 + 2 * 300])
def test_truncated_streams(cut):
    sym = synth.mrz_symbols(3)
    for args in (["--softin", "--ptu"], ["--softin", "-r"], ["--softin", "-R"]):
        _both(args, _soft(sym, sigma=0.2, cut=cut))


def test_noise_only_and_empty():
    rng = np.random.default_rng(11)
    _both(["--softin"], rng.normal(0, 1, 40000).astype(np.float32).tobytes())
    assert _both(["--softin", "-v"], b"") == b""


def test_bad_options():
    assert subprocess.run([BIN, "--nonsense"], capture_output=True).returncode == 255
    assert subprocess.run([BIN, "--br"], capture_output=True).returncode == 255
    r = subprocess.run([BIN, "-", "48000", "16"], input=b"", capture_output=True)
    assert r.returncode == 255 and b"raw data not IQ" in r.stderr
