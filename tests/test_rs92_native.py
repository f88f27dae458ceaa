"""Vaisala RS92 bit-rate tier (include/sonde_rs92.h, host/rs92mod.c --softin / --rawhex): stdout byte for byte against the compiled reference
(`oracle/_ref/rs92mod`) on the same float32 soft-symbol streams and hex lines.  The sonde sends raw GPS ranges; the position comes out of the
decoder's own solver with orbits from a RINEX navigation file or an SEM almanac, so the comparison covers the orbit model, the closed-form /
Bancroft / linearised solutions, DOP, velocity from the delta chips, the leave-one-out search for a bad satellite, the PRN-32 overflow
rules, RS92-NGP keys, calibration and PTU.  Frames, constellation and orbit files from tools/synth_rs92.py; the reference decoding them to
the generator's position pins the generator.  No GPU involved."""
import json
import os
import subprocess

import numpy as np
import pytest

from tools import synth_rs92 as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "rs92mod")
BIN = os.path.join(ROOT, "host", "bin", "rs92mod")

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference not present (oracle/Makefile builds it where /root/reference exists)")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "radiosonde_auto_rx_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])


@pytest.fixture(scope="module")
def orbits(tmp_path_factory):
    d = tmp_path_factory.mktemp("rs92")
    eph = R.constellation()
    E, A = str(d / "brdc.nav"), str(d / "alm.sem")
    open(E, "wb").write(R.rinex_nav(eph, extra_toe=(-7200.0,)))        # two entries per satellite: the nearer one is picked
    open(A, "wb").write(R.sem_almanac(eph, 2100))
    return dict(eph=eph, E=E, A=A, dir=d, flight=R.flight(40, eph))     # 40 frames: all 32 calibration rows come by


def _soft(frames, sigma=0.0, seed=1, invert=False, cut=None):
    sym = R.onair_symbols(frames)
    rng = np.random.default_rng(seed)
    s = 2.0 * sym - 1.0 + rng.normal(0.0, sigma, len(sym))
    if invert:
        s = -s
    if cut is not None:
        s = s[:cut]
    return s.astype(np.float32).tobytes()


def _both(args, data=None):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([BIN] + args, input=data, capture_output=True, timeout=300, env=env)
    b = subprocess.run([REF] + args, input=data, capture_output=True, timeout=300)
    assert a.returncode == b.returncode, (args, a.stderr[-300:], b.stderr[-300:])
    if a.stdout != b.stdout:
        al, bl = a.stdout.splitlines(), b.stdout.splitlines()
        k = next((i for i, (x, y) in enumerate(zip(al, bl)) if x != y), min(len(al), len(bl)))
        raise AssertionError((args, k, al[k:k + 1], bl[k:k + 1], len(al), len(bl)))
    return a.stdout


AUTORX = ["-vx", "-v", "--crc", "--ecc", "--vel", "--json"]                     # decode.py:484,985


def test_rs92_position_is_the_generators(orbits):
    out = _both(AUTORX + ["--ptu", "-e", orbits["E"], "--softin"], _soft(orbits["flight"])).decode()
    js = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(js) == 40 and js[0]["type"] == "RS92" and js[0]["id"] == "K1234567" and js[0]["subtype"] == "RS92-SGP"
    for k, j in enumerate(js):
        assert j["frame"] == 2000 + k and j["datetime"].startswith("2020-04-08T12:00:")
        assert abs(j["lat"] - (47.7123 - 7.25 * k / 111120.0)) < 2e-4 and abs(j["lon"] - (8.9456 + 12.5 * k / (111120.0 * 0.6728))) < 3e-4
        assert abs(j["alt"] - (14321.0 + 5.1 * k)) < 25.0
        assert abs(j["vel_h"] - 14.45) < 0.5 and abs(j["heading"] - 120.1) < 2.0 and abs(j["vel_v"] - 5.1) < 0.5
    assert js[-1]["tx_frequency"] == 402500 and -40.0 < js[-1]["temp"] < -20.0 and 15.0 < js[-1]["humidity"] < 45.0 and 500.0 < js[-1]["pressure"] < 800.0
    assert "temp" not in js[5]                                                       # before the 32 calibration rows are in


@pytest.mark.parametrize("args", [
    [], ["-v"], ["-g1"], ["-g2", "--vel2", "-v"], ["--vel1", "-v"], ["--vel1", "--iter", "-v"], ["-gg", "--vel"], ["-gg", "--vel1"],
    ["-gg", "--vel2", "--iter"], ["-vv", "-vx"], ["--ecc2"], ["--json", "--ecc", "--jsn_cfq", "402500000"], ["--dop", "2.5", "-gg"],
    ["--der", "3", "-g2"], ["--exsat", "17", "-v"], ["--dbg", "--ptu"], ["-r"], ["-r", "-v"]], ids=lambda a: " ".join(a) or "plain")
def test_rs92_options_with_ephemerides(orbits, args):
    _both(args + ["-e", orbits["E"], "--softin"], _soft(orbits["flight"][:34]))


@pytest.mark.parametrize("args", [["--gpsepoch", "2", "-v", "--vel"], ["-g2", "--vel2", "-v"], ["-gg", "--vel1"], ["--json", "--gpsepoch", "2"]], ids=lambda a: " ".join(a))
def test_rs92_almanac(orbits, args):
    """orbits from the almanac: no harmonic terms, positions kilometres off -> the 4000 m limit and the search for the satellite to leave out"""
    d = _soft(orbits["flight"][:12])
    _both(args + ["-a", orbits["A"], "--softin"], d)
    _both(args + ["-a", orbits["A"], "-e", orbits["E"], "--softin"], d)           # ephemerides win
    simple = R.constellation(seed=21, simple=True)                               # a constellation an almanac describes exactly
    p = str(orbits["dir"] / "simple.sem")
    open(p, "wb").write(R.sem_almanac(simple, 2100))
    out = _both(args + ["-a", p, "--softin"], _soft(R.flight(6, simple)))
    assert b"lat: 47.71" in out or b'"lat": 47.71' in out


def test_rs92_noise_inversion_and_short_streams(orbits):
    fr = orbits["flight"]
    P = AUTORX + ["--ptu", "-e", orbits["E"]]
    for sigma, seed in ((0.3, 2), (0.4, 3), (0.5, 4), (0.55, 5), (0.6, 6), (0.65, 7)):                # into and past what RS(255,231) corrects
        _both(P + ["--softin"], _soft(fr, sigma=sigma, seed=seed))
    assert _both(P + ["--softin", "-i"], _soft(fr, invert=True)).count(b'"type": "RS92"') == 40       # the auto_rx form: fsk_demod's symbols are inverted
    assert _both(P + ["--softinv"], _soft(fr, invert=True)).count(b'"type": "RS92"') == 40
    assert _both(P + ["--softin"], _soft(fr, invert=True)) == b""                                     # header of the other polarity: not decoded
    for cut in (30000, 4801 * 3, 200 + 4800 + 60, 200 + 4800 + 61, 1000):
        _both(P + ["--softin"], _soft(fr, cut=cut))
    _both(P + ["--softin"], b"")


def test_rs92_bad_satellite_and_few_satellites(orbits):
    eph, E = orbits["eph"], orbits["E"]
    one = _soft(R.flight(10, eph, spoil={17: 30000.0}))
    for a in (["-g2", "--vel2", "-v"], ["-gg", "--vel"], ["-v", "--vel"], ["-g2", "--vel1", "--iter", "-v"], ["--der", "20000", "-g2", "-v"]):
        _both(a + ["-e", E, "--softin"], one)
    two = _soft(R.flight(10, eph, spoil={17: 3000.0, 28: -2500.0}))
    _both(["-g2", "--vel2", "-v", "-e", E, "--softin"], two)
    _both(["-g2", "--vel2", "-v", "-a", orbits["A"], "--softin"], two)
    for elev in (60.0, 50.0, 40.0):                                                                  # 3, 4, 5 satellites
        few = _soft(R.flight(4, eph, min_elev_deg=elev))
        _both(["-v", "--vel", "--json", "-e", E, "--softin"], few)
        _both(["-g2", "--vel2", "-v", "-e", E, "--softin"], few)


@pytest.mark.parametrize("order", [[32, 17, 28, 6, 11, 1, 13, 19, 24, 30, 3, 9], [17, 28, 32, 6, 11, 1, 13, 19, 24, 30, 3, 9], [17, 32, 28, 6, 11, 1, 13, 19, 24, 30, 3, 9],
                                   [17, 32, 11, 6, 28, 1, 13, 19, 24, 30, 3, 9], [6, 17, 28, 11, 32, 19, 1, 13, 24, 30, 3, 9], [6, 17, 28, 11, 19, 1, 13, 24, 30, 3, 9, 32]],
                         ids=lambda o: "-".join(map(str, o[:6])))
def test_rs92_prn32(orbits, order):
    """PRN 32 does not fit five bits: sent as 0 with the bit above it set — the next number's lowest bit (odd / even neighbours) or the word's spare bit"""
    d = _soft(R.flight(5, orbits["eph"], order=order, min_elev_deg=-90.0))
    _both(["-gg", "--vel", "-e", orbits["E"], "--softin"], d)
    _both(["-g2", "--vel2", "-v", "-e", orbits["E"], "--softin"], d)


def test_rs92_ngp_aux_and_a_second_sonde(orbits):
    eph, E = orbits["eph"], orbits["E"]
    P = ["-vx", "-v", "--vel", "--json", "--ptu", "-e", E, "--softin"]
    caln = R.cal_rows(seed=5, freq_khz=1680500, ngp_key=bytes(range(0x31, 0x41)))
    ngp = _soft(R.flight(40, eph, cal=caln, ngp=True, aux=(0x1234, 0, 0xBEEF, 7)))
    out = _both(P, ngp)                                                                              # the calibration rows switch the type
    assert b'"subtype": "RS92-NGP"' in out and b'"aux": "12340000beef0007"' in out and b'"temp": -2' in out
    assert b'"tx_frequency": 1680500' in _both(["--ngp"] + P, ngp)                                   # (row 0 came by before the type was known above)
    _both(["--ngp", "--dbg", "--ptu", "--softin"], ngp)
    sgp = _soft(R.flight(40, eph, aux=(1, 2, 3, 0xFFFF)))
    _both(["--ngp"] + P, sgp)                                                                        # ... and back
    _both(["-vv", "-vx", "--ptu", "--softin"], sgp)
    cal2 = R.cal_rows(seed=9, freq_khz=404000, killtimer=3000)
    b = [R.rs92_frame(k, cal2, f[72:194], sonde_id="M7654321", frame0=77) for k, f in enumerate(R.flight(36, eph))]
    out = _both(["-vv", "--ptu", "--json", "-e", E, "--softin"], _soft(orbits["flight"][:36] + b))   # another id: the calibration starts over
    assert b"KT:3000s" in out and b'"id": "M7654321"' in out


def test_rs92_week_rollover(orbits):
    """time of week just behind the start of the week with orbit data from the end of the last one, and the other way round"""
    for toe, tow_ms in ((604800.0 - 3600.0, 100_000), (3600.0, 604_700_000)):
        eph = R.constellation(seed=11, toe=toe)
        E, A = str(orbits["dir"] / "roll.nav"), str(orbits["dir"] / "roll.sem")
        open(E, "wb").write(R.rinex_nav(eph))
        open(A, "wb").write(R.sem_almanac(eph, 2100))
        d = _soft(R.flight(4, eph, tow_ms=tow_ms))
        assert _both(["-v", "--vel", "--json", "-e", E, "--softin"], d).count(b'"lat": 47.71') == 4
        _both(["-v", "--vel", "-a", A, "--gpsepoch", "2", "--softin"], d)


def test_rs92_orbit_files_that_are_not(orbits):
    d = _soft(orbits["flight"][:3])
    for a in (["-e", str(orbits["dir"] / "nope.nav")], ["-a", str(orbits["dir"] / "nope.sem")], ["-e", orbits["A"]], ["-a", orbits["E"]]):
        assert _both(["-v"] + a + ["--softin"], d).count(b"lat:") == 0


def test_rs92_rawhex(orbits):
    fr = orbits["flight"]
    hx = "".join(f.hex() + "\n" for f in fr[:6]) + fr[6].hex() + "  [OK] extra\n" + fr[7][:100].hex() + "\n" + fr[8][:60].hex() + "\n" + "zz" * 240 + "\n" + fr[9].hex()
    assert _both(["--rawhex", "-v", "--vel", "--json", "-e", orbits["E"]], hx.encode()).count(b'"type": "RS92"') >= 8
    _both(["--rawhex", "-r", "-v"], hx.encode())
