"""Meteosis MTS01 bit-rate tier (include/sonde_mts01.h, host/mts01mod.c --softin): stdout byte for byte against the compiled reference
(`oracle/_ref/mts01mod`) on the same float32 soft-bit streams.  Frames from tools/synth.py (ASCII telemetry + CRC); the reference printing
[OK] for them pins the generator.  No GPU involved."""
import json
import os
import subprocess

import numpy as np
import pytest

from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "mts01mod")
BIN = os.path.join(ROOT, "host", "bin", "mts01mod")

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


OPTS = [["--softin"], ["--softin", "-v"], ["--softin", "--json"], ["--softin", "-v", "--json", "--jsn_cfq", "402500000"], ["--softin", "-r"], ["--softin", "-R"]]


@pytest.mark.parametrize("args", OPTS, ids=lambda a: " ".join(a))
def test_mts01_clean_and_noisy(args):
    bits = synth.mts01_onair_bits(6)
    out = _both(args, _soft(bits))
    if "-R" not in args:
        assert out.count(b"OK]") == 6
    for sigma, seed in ((0.3, 2), (0.45, 3), (0.6, 4)):
        _both(args, _soft(bits, sigma=sigma, seed=seed))


def test_mts01_fields():
    out = _both(["--softin", "-v", "--json"], _soft(synth.mts01_onair_bits(4), sigma=0.1)).decode()
    js = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(js) == 4
    d = js[2]
    assert d["type"] == "MTS01" and d["id"] == "MTS01-A2031234" and d["frame"] == 3 and d["datetime"] == "2024-06-15T12:00:02.000Z"
    assert abs(d["lat"] - 39.912365) < 2e-5 and abs(d["batt"] - 7.41) < 1e-9 and abs(d["heading"] - 125.4) < 1e-9 and d["version"] == "oracle"
    assert "temp" in d and -20 < d["temp"] < 10


def test_inverted_stream_is_read_raw():
    """the reference does not flip the bits of a header found with negative score (mts01mod.c:580-582,604): the frame comes out complemented"""
    bits = synth.mts01_onair_bits(3)
    a = _both(["--softin", "-r"], _soft(bits, invert=True))
    assert a.count(b"[NO]") == 3
    assert _both(["--softinv", "-r"], _soft(bits, invert=True)).count(b"[OK]") == 3


@pytest.mark.parametrize("cut", [100, 136 + 32 + 8 * 128, 136 + 32 + 8 * 129 + 3, 136 + 32 + 1047, 2 * 1216 + 136 + 32 + 8 * 130])
def test_truncated_streams(cut):
    """a last frame of at least 129 bytes is printed, completed by what the previous frame left in the bit buffer"""
    bits = synth.mts01_onair_bits(4)
    for args in (["--softin"], ["--softin", "-r"], ["--softin", "-R"], ["--softin", "-v", "--json"]):
        _both(args, _soft(bits, sigma=0.2, cut=cut))


def test_noise_only_and_empty():
    rng = np.random.default_rng(11)
    _both(["--softin"], rng.normal(0, 1, 40000).astype(np.float32).tobytes())
    assert _both(["--softin", "-v"], b"") == b""


def test_file_argument_ends_the_argument_list(tmp_path):
    p = tmp_path / "soft.f32"
    p.write_bytes(_soft(synth.mts01_onair_bits(3), sigma=0.1))
    out = _both(["--softin", str(p), "--json"])
    assert out.count(b"[OK]") == 3 and b"{" not in out


def test_bad_options():
    assert subprocess.run([BIN, "--nonsense"], capture_output=True).returncode == 255
    assert subprocess.run([BIN, "--ths"], capture_output=True).returncode == 255
    r = subprocess.run([BIN, "-", "48000", "16"], input=b"", capture_output=True)
    assert r.returncode == 255 and b"raw data not IQ" in r.stderr
