"""RS41 telemetry text / JSON (the reference's print_position(), rs41mod.c:2126-2470) — include/sonde_rs41.h, host side.

Frame streams at the soft-bit level (`rs41mod --softin`, no samples, no GPU): a full calibration cycle with pressure sensor,
a sonde without one plus an ID change, 518-byte frames with xdata, the newer 0x8226 / 0x8329 block layout, frames with broken
block CRCs under a passing ECC, frames beyond the ECC (one / the other / both codewords), and the random-payload frames of the
other fixtures — each through nine option sets (-v, --ptu, --ptu2 --dewp, --json with --jsnsubfrm1/2, --jsn_cfq, --silent,
-r --json, --ecc).  Golden = stdout of the compiled reference on the same soft bits (tools/make_golden.py gen_fields);
the comparison is byte for byte: every printed digit of lat / lon / alt / velocities / T / RH / P / dew point included."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402


@pytest.fixture(scope="module")
def built():
    from radiosonde_auto_rx_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    return np.load(os.path.join(ROOT, "tests", "golden", "rs41_fields.npz"))


@pytest.mark.parametrize("name", sorted(make_golden.FIELD_SCENARIOS))
def test_cli_telemetry_matches_reference(built, name):
    soft = make_golden.fields_softbits(make_golden.FIELD_SCENARIOS[name]).tobytes()
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")          # the reference fixture was built with -DVER_JSN_STR="oracle"
    total = 0
    for k, args in enumerate(make_golden.FIELD_ARGS):
        r = subprocess.run([os.path.join(ROOT, "host", "bin", "rs41mod")] + args + ["--softin"], input=soft, capture_output=True, env=env, timeout=120)
        want = built["%s|%d" % (name, k)].tobytes()
        assert r.returncode == 0
        assert r.stdout == want, (name, args)
        total += len(want)
    assert total > 3000


def test_decoder_api_fields():
    """C ABI through ctypes: frames in, text out, numeric fields readable; option validation."""
    import ctypes as C
    from radiosonde_auto_rx_amd import engine
    from tools import synth
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    L = C.CDLL(engine.LIB_PATH)

    class Opts(C.Structure):
        _fields_ = [(n, C.c_int32) for n in ("verbose", "ptu", "dewp", "json", "jsn_subfrm", "silent", "jsn_freq_khz")] + \
                   [("version", C.c_char * 32), ("reserved", C.c_int32 * 4)]

    class Fields(C.Structure):
        _fields_ = [("frame_nr", C.c_int32), ("id", C.c_char * 12)] + [(n, C.c_int32) for n in ("year", "month", "day", "hour", "minute")] + \
                   [("second", C.c_float), ("is_utc", C.c_int32)] + [(n, C.c_double) for n in ("lat", "lon", "alt", "vel_h", "heading", "vel_v")] + \
                   [("sats", C.c_int32), ("batt", C.c_float), ("temp", C.c_float), ("humidity", C.c_float), ("pressure", C.c_float),
                    ("subtype", C.c_char * 12), ("tx_freq_khz", C.c_int32), ("crc_fail_mask", C.c_int32), ("have_id", C.c_int32),
                    ("have_time", C.c_int32), ("have_pos", C.c_int32)]
    L.sonde_rs41_dec_create.argtypes = [C.POINTER(Opts), C.POINTER(C.c_void_p)]
    L.sonde_rs41_dec_frame.argtypes = [C.c_void_p, C.POINTER(engine.SondeFrame), C.c_char_p, C.c_size_t]
    L.sonde_rs41_dec_fields.argtypes = [C.c_void_p, C.POINTER(Fields)]
    L.sonde_rs41_dec_destroy.argtypes = [C.c_void_p]
    h = C.c_void_p()
    assert L.sonde_rs41_dec_create(C.byref(Opts(verbose=5)), C.byref(h)) < 0            # -vv and beyond: not implemented, refused
    assert L.sonde_rs41_dec_create(C.byref(Opts(ptu=2, json=1, version=b"t")), C.byref(h)) == 0
    fr = synth.rs41_frame(4321, "U1112223", ecef_cm=(418833319, 85974133, 473346430))
    f = engine.SondeFrame(channel=0, len=320, ecc=0, mv_pos=0, mv=1.0, nbytes=518)
    C.memmove(f.frame, fr + bytes(518 - len(fr)), 518)
    buf = C.create_string_buffer(8192)
    n = L.sonde_rs41_dec_frame(h, C.byref(f), buf, 8192)
    text = buf.value.decode()
    assert n == len(text) and text.startswith("[ 4321] (U1112223) ") and '"frame": 4321' in text and '"version": "t"' in text
    fl = Fields()
    assert L.sonde_rs41_dec_fields(h, C.byref(fl)) == 0
    assert fl.frame_nr == 4321 and fl.id == b"U1112223" and abs(fl.lat - 48.1) < 1e-5 and abs(fl.lon - 11.6) < 1e-5 and abs(fl.alt - 12300) < 0.01
    assert fl.have_id and fl.have_time and fl.have_pos and fl.sats == 9 and fl.subtype == b"RS41"
    assert L.sonde_rs41_dec_frame(h, C.byref(f), buf, 10) < 0                            # output does not fit: error, nothing truncated
    L.sonde_rs41_dec_destroy(h)
