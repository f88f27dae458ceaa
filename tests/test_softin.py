"""Soft-bit input framing (`rs41mod --softin -i`, the consumer of `fsk_demod -s` in auto_rx's decode.py:901-909).

Host-side bit-rate logic of libsonde_hip (sonde_softin_*), no GPU involved: the reference's fsk_demod soft decisions stored
in the modem fixtures must frame and decode to exactly the lines the reference's own `rs41mod --softin -i -r --ecc2` printed."""
import os
import subprocess

import numpy as np
import pytest
from golden_cases import load_fsk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["fsk_rs41_48k_mask", "fsk_rs41_48k_peak"])
def test_softin_cli_matches_reference_lines(name):
    from radiosonde_auto_rx_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    g = load_fsk(name)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", "rs41mod"), "--softin", "-i", "-r", "--ecc2"],
                       input=g["sd"].astype("<f4").tobytes(), capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.decode().splitlines() == g["rs41_lines"]
    # wrong polarity without -i: the header correlates negatively and is rejected (rs41mod.c:2888-2891)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", "rs41mod"), "--softin", "-r", "--ecc2"],
                       input=g["sd"].astype("<f4").tobytes(), capture_output=True, timeout=60)
    assert r.stdout == b""
    # --softinv negates the stream instead
    r = subprocess.run([os.path.join(ROOT, "host", "bin", "rs41mod"), "--softinv", "-r", "--ecc2"],
                       input=g["sd"].astype("<f4").tobytes(), capture_output=True, timeout=60)
    assert r.stdout.decode().splitlines() == g["rs41_lines"]


def test_softin_truncated_stream_emits_partial_frame():
    """EOF inside a frame: the reference prints the frame with the bytes read so far (rs41mod.c:2931,2965)."""
    g = load_fsk("fsk_rs41_48k_mask")
    sd = g["sd"].astype("<f4").ravel()
    ref = os.path.join(ROOT, "oracle", "_ref", "rs41mod")
    if not os.path.exists(ref):
        pytest.skip("compiled reference not present")
    cut = sd[:len(sd) * 55 // 100]
    a = subprocess.run([os.path.join(ROOT, "host", "bin", "rs41mod"), "--softin", "-i", "-r", "--ecc2"], input=cut.tobytes(), capture_output=True)
    b = subprocess.run([ref, "--softin", "-i", "-r", "--ecc2"], input=cut.tobytes(), capture_output=True)
    assert a.stdout == b.stdout and a.stdout


def test_dfm_softin_cli_matches_reference_lines():
    """`dfm09mod --softin [-i] -r --ecc`: two soft symbols per bit, 8 frames per header hit (dfm09mod.c:1604-1720)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    g = load_fsk("fsk_dfm_50k")
    sd = g["sd"].astype("<f4").tobytes()
    for key, args in (("dfm_lines", ["--softin", "-i", "-r", "--ecc"]), ("dfm_lines_noinv", ["--softin", "-r", "--ecc"])):
        r = subprocess.run([os.path.join(ROOT, "host", "bin", "dfm09mod")] + args, input=sd, capture_output=True, timeout=60)
        assert r.returncode == 0, r.stderr
        assert [l.rstrip() for l in r.stdout.decode().splitlines()] == [str(l).rstrip() for l in g[key]] and len(g[key]) > 0


def test_bin_cli_matches_reference_lines():
    """`rs41mod|dfm09mod --bin [-i|--auto] -r`: one byte per hard bit (fsk_demod without -s), header found by bit errors in either
    polarity (find_binhead / cmp_hdb, demod_mod.c:1639-1690).  Golden = the reference decoders on the same bytes."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    from radiosonde_auto_rx_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    gold = np.load(os.path.join(ROOT, "tests", "golden", "bin_lines.npz"))
    some = 0
    for name, binary, extra, flags in make_golden.BIN_RUNS:
        bits = (load_fsk(name)["sd"].ravel() < 0).astype(np.uint8).tobytes()
        r = subprocess.run([os.path.join(ROOT, "host", "bin", binary), "--bin", "-r"] + extra + flags, input=bits, capture_output=True, timeout=60)
        want = [str(l).rstrip() for l in gold["|".join([name, binary] + flags)]]
        assert r.returncode == 0 and [l.rstrip() for l in r.stdout.decode().splitlines()] == want, (name, flags)
        some += len(want)
    assert some > 10


def test_rawhex_cli_matches_reference_lines():
    """`rs41mod --rawhex -r [--ecc|--ecc2]`: frames as hex lines (rs41mod.c:2976-3002) — clean, correctable, uncorrectable,
    too short (skipped) and truncated lines; the frame buffer persists between lines like the reference's."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    g = np.load(os.path.join(ROOT, "tests", "golden", "rawhex_lines.npz"))
    for key, flags in (("ecc2", ["--ecc2"]), ("ecc", ["--ecc"]), ("none", [])):
        r = subprocess.run([os.path.join(ROOT, "host", "bin", "rs41mod"), "--rawhex", "-r"] + flags, input=str(g["input"]).encode(),
                           capture_output=True, timeout=60)
        assert r.returncode == 0
        assert r.stdout.decode().splitlines() == [str(l) for l in g[key]] and len(g[key]) == 5


def _ref_or_skip(name):
    p = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(p):
        pytest.skip("compiled reference not present")
    return p


def test_rs41_sat_and_argument_order_match_reference():
    """`--sat` (raw GPS block contents incl. the newer GNSS block, no PTU then: rs41mod.c:2052-2111,1221-1260,2279), `-vx` / `-vv` (xdata text, battery,
    week, sats, subframe bytes, QFE: :1565-1578,1981,2018,2029,1492-1506), `--aux` (ECC / OIF411 / CFH instrument records in the xdata text, :1280-1452) and the order dependence of `--json` / `--ecc` (`--json` sets ecc = 2
    where it stands, a later `--ecc` wins, :2703-2707; `--jsnsubfrm1` forces 2 afterwards, :2769-2773) — found by tests/fuzz/fuzz_family.py"""
    import sys
    sys.path.insert(0, ROOT)
    from tools import synth
    ref = _ref_or_skip("rs41mod")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    rng = np.random.default_rng(5)
    cal = synth.rs41_cal_table(3)
    bits = np.concatenate([synth.rs41_onair_bits(synth.rs41_frame(k, cal_table=cal, ptu_counts=True, xdata=[["0103123407D02D78"], ["05011B5801F4A782640A", "0501N1234567003C0123I"], ["08011234567890ABCDEF1234", "0102", "zz"]][k % 3] if k % 2 else None, gnss2=(k > 33),
                                                                  ecef_cm=(418833319, 85974133, 473346430))) for k in (0, 1, 2, 33, 34, 49, 50)])
    clean = (2.0 * bits - 1.0).astype(np.float32)
    hurt = clean.copy()
    pos = rng.choice(np.arange(2000, len(hurt)), 55, replace=False)          # more byte errors than --ecc corrects in one of the code words
    hurt[pos] *= -1
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    for args in (["--softin", "--sat", "-v", "--ecc2"], ["--softin", "--sat", "--ptu", "--ecc"], ["--softin", "--sat", "--silent", "--ecc2"],
                 ["--softin", "-vv", "--ecc2", "--ptu"], ["--softin", "-vx", "--ecc2"], ["--softin", "-vv", "--sat", "--ecc2"], ["--softin", "--aux", "--ecc2"],
                 ["--softin", "--aux", "--json", "--ptu"],
                 ["--softin", "--json", "--ecc"], ["--softin", "--ecc", "--json"], ["--softin", "--ecc3", "--jsnsubfrm1"], ["--softin", "--json", "--ecc3"]):
        for data in (clean, hurt):
            a = subprocess.run([os.path.join(ROOT, "host", "bin", "rs41mod")] + args, input=data.tobytes(), capture_output=True, timeout=60, env=env)
            b = subprocess.run([ref] + args, input=data.tobytes(), capture_output=True, timeout=60)
            assert a.returncode == b.returncode == 0 and a.stdout == b.stdout, (args, a.stdout[:300], b.stdout[:300])
    out = subprocess.run([os.path.join(ROOT, "host", "bin", "rs41mod"), "--softin", "--sat", "-v", "--ecc2"], input=clean.tobytes(), capture_output=True).stdout
    assert out.count(b"iTOW: 0x") >= 2 and out.count(b"ECEF-POS: (") >= 2 and b"prMes:" in out and b"numSV168" in out


@pytest.mark.parametrize("dec,opts", [("rs41mod", ["-r", "--ecc2"]), ("rs41mod", ["--xorhex", "-r", "--ecc"]), ("m10mod", ["-r", "-v"]), ("m20mod", ["-vv"])])
def test_rawhex_lines_with_non_hex_characters(dec, opts):
    """a pair of characters that is not hex keeps the previous byte (the reference's sscanf leaves its variable alone, rs41mod.c:2995, m10mod.c:1539,
    m20mod.c:1405), also across lines; odd lengths, blanks, text behind the frame — found by tests/fuzz/fuzz_family.py"""
    import sys
    sys.path.insert(0, ROOT)
    from tools import synth
    ref = _ref_or_skip(dec)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    if dec == "rs41mod":
        h = synth.rs41_frame(7).hex()
    else:
        h = (synth.m10_frame(3, rng=np.random.default_rng(3)) if dec == "m10mod" else synth.m20_frame(3)).hex()
    lines = [h, h[:60] + "zz" + h[62:], h[:61], "xy" + h[2:200] + " [OK] tail", h[:100] + "g" + h[101:] + "\n" + "q" * 80, h[:2 * 40] + "  " + h[2 * 41:], "", "#" * 300]
    data = ("\n".join(lines) + "\n").encode()
    args = ([] if "--xorhex" in opts else ["--rawhex"]) + opts
    a = subprocess.run([os.path.join(ROOT, "host", "bin", dec)] + args, input=data, capture_output=True, timeout=60)
    b = subprocess.run([ref] + args, input=data, capture_output=True, timeout=60)
    assert a.returncode == b.returncode == 0 and a.stdout == b.stdout and a.stdout, (a.stdout[:300], b.stdout[:300])


def test_dfm_rawecc_and_packet_hex_match_reference():
    """`dfm09mod --rawecc` (the frame's bits before the Hamming decoder as hex — what auto_rx asks for when it saves raw frames, decode.py:1078) and
    `-R` (the nine data packets as hex, dfm09mod.c:972-981) on soft symbols, clean and damaged"""
    ref = _ref_or_skip("dfm09mod")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    sd = load_fsk("fsk_dfm_50k")["sd"].astype("<f4").ravel()
    rng = np.random.default_rng(9)
    hurt = sd + rng.normal(0, 0.6 * np.abs(sd).mean(), len(sd)).astype(np.float32)
    some = 0
    for args in (["--softin", "-i", "--rawecc"], ["-vv", "--ecc", "--json", "--dist", "--auto", "--softin", "--rawecc"], ["--softin", "-i", "-R", "--ecc"], ["--softin", "-R"],
                 ["--softin", "--auto", "--rawecc", "--ecc2", "--json"]):
        for data in (sd, hurt, sd[:len(sd) // 3]):
            a = subprocess.run([os.path.join(ROOT, "host", "bin", "dfm09mod")] + args, input=data.tobytes(), capture_output=True, timeout=60, env=dict(os.environ, SONDE_JSN_VERSION="oracle"))
            b = subprocess.run([ref] + args, input=data.tobytes(), capture_output=True, timeout=60)
            assert a.returncode == b.returncode == 0 and a.stdout == b.stdout, (args, a.stdout[:300], b.stdout[:300])
            some += len(a.stdout)
    assert some > 2000
    # --rawhex: the --rawecc text back in (dfm09mod.c:1730-1787), also with blanks, other characters, wrong lengths
    import sys
    sys.path.insert(0, ROOT)
    from tools import synth
    bits = []
    for k in range(24):
        d1 = [int(v) for v in rng.integers(0, 16, 13)]; d1[12] = k % 9
        d2 = [int(v) for v in rng.integers(0, 16, 13)]; d2[12] = (k + 4) % 9
        bits.append(synth.dfm_frame_bits([int(v) for v in rng.integers(0, 16, 7)], d1, d2))
    b = np.concatenate(bits)
    sym = np.empty(2 * len(b), np.float32); sym[0::2] = 1.0 - 2.0 * b; sym[1::2] = 2.0 * b - 1.0         # bit 1 -> symbols 0, 1
    sym += rng.normal(0, 0.35, len(sym)).astype(np.float32)
    raw = subprocess.run([ref, "--softin", "--auto", "--rawecc"], input=sym.tobytes(), capture_output=True, timeout=60).stdout
    assert raw.count(b"\n") >= 12
    lines = raw.split(b"\n")
    dirty = b"\n".join(lines[:5] + [lines[5][:40] + b"zz" + lines[5][40:], lines[6][:-3], b"+<12.5>" + lines[7][11:], b"garbage", lines[8].replace(b" ", b"   ")] + lines[9:])
    for args in (["--rawhex", "-vv", "--ecc", "--json", "--dist"], ["--rawhex", "-r", "--ecc"], ["--rawhex", "--rawecc"], ["--rawhex", "-R", "--ecc2"], ["--rawhex", "--ptu"]):
        for data in (raw, dirty):
            a = subprocess.run([os.path.join(ROOT, "host", "bin", "dfm09mod")] + args, input=data, capture_output=True, timeout=60, env=dict(os.environ, SONDE_JSN_VERSION="oracle"))
            b = subprocess.run([ref] + args, input=data, capture_output=True, timeout=60)
            assert a.returncode == b.returncode == 0 and a.stdout == b.stdout, (args, a.stdout[:300], b.stdout[:300])
