"""GPU parity of the M10 demodulator path (`m10mod -r [-v]`): 9615 Bd Manchester GFSK (5 samples per symbol at the 48 kHz IF:
the direct, non-factorised header correlation), 32-symbol raw header compared per symbol, 968 differentially coded bits per
frame, the rest of the second skipped, either polarity (m10mod.c:1370-1390,1436-1510).  Input forms: --IQ (48 kHz and 2.4 Msps,
with --dc), --iq0 / --iq2, FM audio, a stream ending inside a frame, a frame with aux bytes.
Golden = stdout / stderr / exit code of the compiled reference on the same bytes (tools/make_golden.py M10_CASES): the hex lines
with checksum value and verdict must be identical."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


ALL = {**{k: (v, "m10mod") for k, v in make_golden.M10_CASES.items()}, **{k: (v, "m20mod") for k, v in make_golden.M20_CASES.items()}}


@pytest.mark.parametrize("name", sorted(ALL))
def test_cli_m10_matches_reference(name):
    """M10 (m10mod) and M20 (m20mod: 9600 Bd, length byte 0x45 / shorter / longer with aux bytes, block checksum verdict)"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    case, binary = ALL[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    stdin, args = make_golden.m10_capture_cli(case)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", binary)] + args, input=stdin, capture_output=True, timeout=180)
    assert r.returncode == int(g["rc"]), (r.returncode, r.stderr)
    assert r.stderr.decode() == str(g["stderr"])
    want = g["stdout"].tobytes().decode().splitlines()
    assert [l.rstrip() for l in r.stdout.decode().splitlines()] == [l.rstrip() for l in want]
    assert len(want) >= 1


def test_m10_engine_many_channels():
    """Batched form: 6 channels of one engine, each its own capture and carrier; frames per channel equal the single-channel CLI
    goldens' frame bytes for the matching capture."""
    from radiosonde_auto_rx_amd.engine import lib, SondeCfg, ABI_VERSION, _chk
    import ctypes as C
    from radiosonde_auto_rx_amd import synth

    class M10Frame(C.Structure):
        _fields_ = [("channel", C.c_int32), ("nbits", C.c_int32), ("len", C.c_int32), ("cs_ok", C.c_int32), ("cs_calc", C.c_uint32),
                    ("mv_pos", C.c_uint32), ("mv", C.c_float), ("frame", C.c_uint8 * 124)]
    sr = 2_400_000
    fqs = [synth.snap_fq(f, sr) for f in (0.11, -0.2, 0.3, -0.05, 0.01, 0.4)]
    caps = [synth.m10_capture(sr=sr, seconds=1.6, fq=fq, noise_sigma=0.02, seed=20 + k, t_first=0.2 + 0.03 * k) for k, fq in enumerate(fqs)]
    x = np.stack(caps)
    fq_arr = np.array(fqs, np.float64)
    cfg = SondeCfg(abi_version=ABI_VERSION, n_channels=len(fqs), sample_rate=sr, bits=16, sonde_type=10, opt_lp=1, max_chunk=sr, max_frames=32)
    h = C.c_void_p()
    L = lib()
    _chk(L.sonde_engine_create(C.byref(cfg), fq_arr.ctypes.data_as(C.POINTER(C.c_double)), C.byref(h)))
    L.sonde_engine_fetch_m10.argtypes = [C.c_void_p, C.POINTER(M10Frame), C.c_int32, C.c_int32]
    n = x.shape[1] // 2
    got = {}
    buf = (M10Frame * 32)()
    for s0 in range(0, n, sr // 2):
        s1 = min(n, s0 + sr // 2)
        chunk = np.ascontiguousarray(x[:, 2 * s0:2 * s1])
        _chk(L.sonde_engine_process_host(h, chunk.ctypes.data_as(C.c_void_p), s1 - s0, s1 - s0))
        k = _chk(L.sonde_engine_fetch_m10(h, buf, 32, int(s1 >= n)))
        for i in range(k):
            got.setdefault(buf[i].channel, []).append(bytes(buf[i].frame[:buf[i].len]))
    L.sonde_engine_destroy(h)
    assert sorted(got) == list(range(len(fqs)))
    for c in range(len(fqs)):
        assert len(got[c]) >= 1 and all(f[:2] == bytes([0x64, 0x9F]) and len(f) == 101 for f in got[c][:1])


@pytest.mark.parametrize("args", [["--json", "--ptu", "-vvv"], ["-vv", "--ptu"], ["-r", "-v", "--json"]])
def test_m10_telemetry_on_iq_matches_reference(args):
    """Telemetry frames (valid GPS / sensor words / checksum) modulated at 48 kHz: `m10mod <args> --IQ 0.0 --lpIQ - 48000 16` from this repo
    (GPU demodulator + host framer + telemetry tier; -vvv = no skipping behind a frame) against the compiled reference."""
    from radiosonde_auto_rx_amd import synth
    ref = os.path.join(ROOT, "oracle", "_ref", "m10mod")
    if not os.path.exists(ref):
        pytest.skip("compiled reference not present")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    x = synth.m10_capture(sr=48_000, seconds=6.3, noise_sigma=0.02, seed=31, f_offset_hz=250.0,
                          frame_fn=lambda k: synth.m10_frame(k, rng=np.random.default_rng(500 + k), good_checksum=(k != 2)))
    tail = ["--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([os.path.join(ROOT, "host", "bin", "m10mod")] + args + tail, input=x.tobytes(), capture_output=True, timeout=180, env=env)
    b = subprocess.run([ref] + args + tail, input=x.tobytes(), capture_output=True, timeout=180)
    assert a.returncode == 0 and a.stdout == b.stdout
    assert len(a.stdout.splitlines()) >= 5
