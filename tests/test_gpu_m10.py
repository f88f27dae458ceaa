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
    """Batched form: 11 channels of one 2.4 Msps engine (the hand-scheduled decimator with its channel -> XCD mapping), each its own
    capture, carrier, offset, noise and start time, fed from device memory; per channel every text line equals the stdout of the compiled
    reference `m10mod -r -v --IQ fq --lpIQ - 2400000 16` on that capture."""
    import torch
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    ref = os.path.join(ROOT, "oracle", "_ref", "m10mod")
    if not os.path.exists(ref):
        pytest.skip("compiled reference not present")
    sr = 2_400_000
    fqs = [synth.snap_fq(f, sr) for f in (0.11, -0.2, 0.3, -0.05, 0.01, 0.4, -0.33, 0.22, -0.44, 0.17, -0.12)]
    caps = [synth.m10_capture(sr=sr, seconds=2.3, fq=fq, noise_sigma=(0.02, 0.05, 0.1)[k % 3], seed=20 + k, t_first=0.2 + 0.03 * k, f_offset_hz=150.0 * (k - 5),
                              frame_fn=lambda j, k=k: synth.m10_frame(j, rng=np.random.default_rng(40 * k + j), good_checksum=(j + k) % 4 != 3))
            for k, fq in enumerate(fqs)]
    x = np.stack(caps)
    dev = torch.from_numpy(x).to("cuda:0")
    eng = Engine(fqs, sr, sonde="m10", max_chunk=sr, max_frames=64)
    n = x.shape[1] // 2
    n -= n % 50
    got = {}
    for s0 in range(0, n, sr):
        s1 = min(n, s0 + sr)
        eng.process_device(dev.data_ptr() + 4 * s0, x.shape[1] // 2, s1 - s0)
        for f in eng.fetch_mxx(finish=s1 >= n):
            got.setdefault(f["channel"], []).append(f["line"].rstrip())
    eng.close()
    assert sorted(got) == list(range(len(fqs)))
    for c, fq in enumerate(fqs):
        r = subprocess.run([ref, "-r", "-v", "--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"], input=x[c, :2 * n].tobytes(), capture_output=True, timeout=180)
        want = [l.rstrip() for l in r.stdout.decode().splitlines()]
        assert got[c] == want and len(want) >= 2, c


@pytest.mark.parametrize("args", [["--json", "--ptu", "-vvv"], ["-vv", "--ptu"], ["-r", "-v", "--json"]])
def test_m10_telemetry_on_iq_matches_reference(args):
    """Telemetry frames (valid GPS / sensor words / checksum) modulated at 48 kHz: `m10mod <args> --IQ 0.0 --lpIQ - 48000 16` from this repo
    (GPU demodulator + host framer + telemetry tier; -vvv = no skipping behind a frame) against the compiled reference."""
    from tools import synth
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


@pytest.mark.parametrize("args", [["--json", "--ptu", "-vvv"], ["-v", "--ptu"], ["-r", "-v", "--json"]])
def test_m20_telemetry_on_iq_matches_reference(args):
    """M20 telemetry frames (firmware 8 with pressure, one bad checksum) at 9600 Bd: `m20mod <args> --IQ 0.0 --lpIQ - 48000 16`"""
    from tools import synth
    ref = os.path.join(ROOT, "oracle", "_ref", "m20mod")
    if not os.path.exists(ref):
        pytest.skip("compiled reference not present")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    x = synth.m10_capture(sr=48_000, seconds=6.3, noise_sigma=0.02, seed=33, f_offset_hz=-180.0, baud=9600.0,
                          frame_fn=lambda k: synth.m20_frame(k, fw=8, pressure_hpa=640.5 - 30 * k, rng=np.random.default_rng(700 + k),
                                                             good_checksum=(k != 3)))
    tail = ["--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    a = subprocess.run([os.path.join(ROOT, "host", "bin", "m20mod")] + args + tail, input=x.tobytes(), capture_output=True, timeout=180, env=env)
    b = subprocess.run([ref] + args + tail, input=x.tobytes(), capture_output=True, timeout=180)
    assert a.returncode == 0 and a.stdout == b.stdout
    assert len(a.stdout.splitlines()) >= 5


@pytest.mark.parametrize("binary,shift", [("rs41mod", "1"), ("rs41mod", "-3"), ("dfm09mod", "-1"), ("m10mod", "2"), ("m20mod", "1")])
def test_cli_bit_offset_option_matches_reference(binary, shift):
    """-d <shift> of the native front ends (sonde_engine_set_sync behind it) against the compiled reference"""
    from tools import synth
    ref = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(ref):
        pytest.skip("compiled reference not present")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    if binary == "rs41mod":
        x = synth.rs41_capture(sr=48_000, seconds=4.3, fq=0.0, noise_sigma=0.3, n_frames=4, t_first=0.15, seed=61)
        args = ["-r", "--ecc2", "--crc"]
    elif binary == "dfm09mod":
        x = synth.dfm_capture(sr=48_000, seconds=3.2, fq=0.0, noise_sigma=0.15, seed=62)
        args = ["-r", "--ecc2"]
    else:
        fn = (lambda k: synth.m10_frame(k, rng=np.random.default_rng(80 + k))) if binary == "m10mod" else (lambda k: synth.m20_frame(k, rng=np.random.default_rng(90 + k)))
        x = synth.m10_capture(sr=48_000, seconds=4.3, noise_sigma=0.1, seed=63, baud=9616.0 if binary == "m10mod" else 9600.0, frame_fn=fn)
        args = ["-r", "-v"]
    tail = ["-d", shift, "--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
    a = subprocess.run([os.path.join(ROOT, "host", "bin", binary)] + args + tail, input=x.tobytes(), capture_output=True, timeout=180)
    b = subprocess.run([ref] + args + tail, input=x.tobytes(), capture_output=True, timeout=180)
    assert a.returncode == 0 and a.stdout == b.stdout and len(b.stdout.splitlines()) >= 2


@pytest.mark.parametrize("binary,extra", [("dfm09mod", ["--br", "2497.5"]), ("dfm09mod", ["--br", "3100"]), ("m20mod", ["--br", "9616"]), ("m20mod", ["--br", "12"]),
                                          ("m10mod", ["--chk3"]), ("m10mod", ["--chk3", "--iq2"])])
def test_cli_br_and_chk3_match_reference(binary, extra):
    """`dfm09mod --br x` / `m20mod --br x` (symbol rate replaced before the design: dfm09mod.c:1436-1443,1590-1594; m20mod.c:1082-1089; out of range
    = default) and `m10mod --chk3` (bits from both soft values of read_softbit2p, m10mod.c:1476-1479) on noisy captures whose symbol clock is off
    the nominal one: stdout and the stderr preamble of the compiled reference."""
    from tools import synth
    ref = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(ref):
        pytest.skip("compiled reference not present")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    if binary == "dfm09mod":
        x = synth.dfm_capture(sr=48_000, seconds=4.2, fq=0.0, noise_sigma=0.12, seed=162)
        args = ["-r", "--ecc2"]
    elif binary == "m20mod":
        x = synth.m10_capture(sr=48_000, seconds=4.3, noise_sigma=0.12, seed=163, baud=9616.0, frame_fn=lambda k: synth.m20_frame(k, rng=np.random.default_rng(190 + k)))
        args = ["-r", "-v"]
    else:
        x = synth.m10_capture(sr=48_000, seconds=5.3, noise_sigma=0.15, seed=164, frame_fn=lambda k: synth.m10_frame(k, rng=np.random.default_rng(180 + k)))
        args = ["-r", "-v"]
    tail = ["-", "48000", "16"] if "--iq2" in extra else ["--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
    a = subprocess.run([os.path.join(ROOT, "host", "bin", binary)] + args + extra + tail, input=x.tobytes(), capture_output=True, timeout=180)
    b = subprocess.run([ref] + args + extra + tail, input=x.tobytes(), capture_output=True, timeout=180)
    assert a.returncode == 0 and a.stdout == b.stdout and len(b.stdout.splitlines()) >= 2
    assert a.stderr.decode().splitlines()[:3] == b.stderr.decode().splitlines()[:3]
