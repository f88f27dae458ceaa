"""The function-level seam: the reference's OWN decoders — their main(), option handling, framing, ECC, telemetry
and JSON code, compiled from the sources where they lie — linked against host/seam/demod_mod_hip.c instead of demod_mod.c, so that
init_buffers() / find_header() / read_softbit*() run on the GPU engine.  oracle/Makefile builds them (oracle/_ref/*_seam); here they
run next to the all-CPU reference binaries on the same input and must print the same stdout, byte for byte.

Covers the argv auto_rx uses (decode.py:417,517,544) on --IQ input, FM-audio WAV, IF-rate IQ, --dc, inverted polarity with -i / --auto,
8-bit input, and end of input inside a frame."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402

pytestmark = pytest.mark.gpu
REF = os.path.join(ROOT, "oracle", "_ref")
ECEF = dict(ecef_cm=(418833319, 85974133, 473346430))


def _both(binary, args, stdin):
    seam, ref = os.path.join(REF, binary + "_seam"), os.path.join(REF, binary)
    if not (os.path.exists(seam) and os.path.exists(ref)):
        pytest.skip("oracle/_ref seam binaries not built (make -C oracle ref)")
    a = subprocess.run([seam] + args, input=stdin, capture_output=True, timeout=300)
    b = subprocess.run([ref] + args, input=stdin, capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0, (a.returncode, a.stderr[-400:])
    assert a.stdout == b.stdout, (binary, args, a.stdout[:300], b.stdout[:300])
    return a.stdout


def _rs41(sr, secs, fq, **kw):
    from tools import synth
    fq = synth.snap_fq(fq, sr)
    return synth.rs41_capture(sr=sr, seconds=secs, fq=fq, noise_sigma=0.02, frame_kw=ECEF, **kw), fq


@pytest.mark.parametrize("args", [["--ptu2", "--json", "--jsnsubfrm1"], ["-r", "--ecc2", "--crc"], ["-v", "--ecc", "--dc", "--json"]])
def test_seam_rs41_iq(args):
    x, fq = _rs41(2_400_000, 4.3, 0.12, n_frames=4, t_first=0.1, seed=91, bit_errors=3)
    out = _both("rs41mod", args + ["--IQ", repr(fq), "--lpIQ", "-", "2400000", "16"], x.tobytes())
    assert len(out.splitlines()) >= 4


def test_seam_rs41_eof_inside_frame_and_8bit():
    from tools import synth
    x, fq = _rs41(2_400_000, 3.3, -0.2, n_frames=3, t_first=0.1, seed=92)
    x = x[:2 * int(2_400_000 * 2.4)]                                           # third frame cut by the end of the stream
    out = _both("rs41mod", ["-r", "--ecc2", "--IQ", repr(fq), "--lpIQ", "-", "2400000", "16"], x.tobytes())
    assert len(out.splitlines()) == 3 and out.splitlines()[2].endswith(b"[NO] (--)")
    _both("rs41mod", ["-r", "--ecc2", "--IQ", repr(fq), "--lpIQ", "-", "2400000", "8"], synth.to_u8(x).tobytes())


def test_seam_rs41_audio_wav_and_ifiq():
    from tools import synth
    x, _ = _rs41(48_000, 4.3, 0.0, n_frames=4, t_first=0.15, seed=93)
    out = _both("rs41mod", ["--ptu2", "--json", "--jsnsubfrm1"], synth.wav_bytes(synth.fm_audio(x), 48_000))
    assert out.count(b'"type": "RS41"') >= 3
    _both("rs41mod", ["-r", "--ecc2", "--iq2", "--lpIQ", "-", "48000", "16"], x.tobytes())
    _both("rs41mod", ["-r", "--ecc2", "--iq0", "--iqdc", "-", "48000", "16"], x.tobytes())


@pytest.mark.parametrize("name", ["inv_rs41_2400k_i", "inv_rs41_2400k_auto", "inv_rs41_2400k_i_on_normal", "inv_dfm_2400k_auto", "inv_dfm_2400k_i"])
def test_seam_polarity(name):
    _, stdin, binary, args, _ = make_golden.inv_capture(make_golden.INV_CASES[name])
    _both(binary, args, stdin)


def test_seam_dfm_autorx_args():
    from tools import synth
    sym = (make_golden.dfm_field_symbols(dict(kind="09", n=40, sn=18012345)) > 0).astype(np.uint8)
    sr = 48_000
    z = 0.5 * synth.gfsk_baseband(sym, sr, 2500.0, 2400.0)
    rng = np.random.default_rng(7)
    z = z * np.exp(2j * np.pi * 300.0 / sr * np.arange(len(z))) + 0.02 * (rng.standard_normal(len(z)) + 1j * rng.standard_normal(len(z)))
    x = np.empty(2 * len(z), np.int16)
    x[0::2] = np.round(z.real * 32767 * 0.9); x[1::2] = np.round(z.imag * 32767 * 0.9)
    out = _both("dfm09mod", ["-vv", "--ecc", "--json", "--dist", "--auto", "--IQ", "0.0", "--lpIQ", "-", str(sr), "16"], x.tobytes())
    assert out.count(b'"type": "DFM"') >= 3


@pytest.mark.parametrize("binary,baud", [("m10mod", 9616.0), ("m20mod", 9600.0)])
def test_seam_m10_m20(binary, baud):
    from tools import synth
    fn = (lambda k: synth.m10_frame(k, rng=np.random.default_rng(40 + k))) if binary == "m10mod" else \
         (lambda k: synth.m20_frame(k, fw=8, pressure_hpa=700.0 - k, rng=np.random.default_rng(60 + k)))
    x = synth.m10_capture(sr=48_000, seconds=5.3, noise_sigma=0.02, seed=95, f_offset_hz=200.0, baud=baud, frame_fn=fn)
    out = _both(binary, ["--json", "--ptu", "-vv", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], x.tobytes())
    assert out.count(b'"type"') >= 4
    _both(binary, ["-r", "-v", "--iq2", "-", "48000", "16"], x.tobytes())


@pytest.mark.parametrize("binary", ["rs92mod", "imet54mod", "mp3h1mod", "mts01mod", "meisei100mod"])
def test_seam_family_generic(binary):
    """The frame-based rest of the reference's demod/mod family through the engine's generic sonde description (baud, header, BT, h, symbol
    layout taken from the decoder's own dsp_t; bits per hit from the seam's table): raw output on baseband IQ at 2.4 Msps (mixer +
    decimator), IF-rate IQ with the tone correlator (--iq3: centre window), FM-sliced (--iq0), and an inverted signal."""
    from tools import synth
    sr = 2_400_000
    fq = synth.snap_fq(0.07, sr)
    x = synth.family_capture(binary, sr=sr, seconds=3.4, fq=fq, seed=11, f_offset_hz=150.0)
    out = _both(binary, ["-r", "--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"], x.tobytes())
    assert len(out.splitlines()) >= 2
    y = synth.family_capture(binary, sr=48_000, seconds=4.4, seed=12)
    _both(binary, ["-r", "--iq3", "--lpIQ", "-", "48000", "16"], y.tobytes())
    _both(binary, ["-r", "--iq0", "-", "48000", "16"], y.tobytes())
    _both(binary, ["-r", "--IQ", "0.0", "--lpIQ", "--dc", "-", "48000", "16"], y.tobytes())
    z = synth.family_capture(binary, sr=48_000, seconds=3.4, seed=13, invert=True)
    _both(binary, ["-r", "-i", "--IQ", "0.0", "-", "48000", "16"], z.tobytes())


def test_seam_lms6():
    """lms6Xmod (LMS6-403: RS(255,223) blocks behind a K = 7 rate-1/2 convolutional code; 64-raw-bit header, up to 10 header errors accepted,
    4096 raw bits per hit) on the seam: told apart from RS41 — same 4800 Bd / 64-bit header — by the header itself; `--lmsX` sets dsp.br = 4797.8
    AFTER init_buffers() (lms6Xmod.c:1343-1347), which the seam picks up for the bit clock at the first find_header()."""
    from tools import synth
    sr = 2_400_000
    fq = synth.snap_fq(-0.06, sr)
    x = synth.lms6_capture(sr=sr, seconds=4.0, fq=fq, noise_sigma=0.05, seed=21)
    tail = ["--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"]
    out = _both("lms6Xmod", ["--vit", "--ecc", "--json"] + tail, x.tobytes())
    assert out.count(b'"type": "LMS"') >= 3 and out.count(b"[OK]") >= 3
    y = synth.lms6_capture(sr=48_000, seconds=5.0, noise_sigma=0.12, seed=22)
    raw = _both("lms6Xmod", ["-r", "--ecc", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], y.tobytes())
    assert raw.count(b"[OK]") >= 3
    _both("lms6Xmod", ["--lms6", "--vit2", "--ecc3", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], y.tobytes())     # soft Viterbi: both soft bits
    _both("lms6Xmod", ["-r", "--iq3", "--lpIQ", "-", "48000", "16"], y.tobytes())
    z = synth.lms6_capture(sr=48_000, seconds=5.0, noise_sigma=0.01, seed=23)
    assert _both("lms6Xmod", ["-r", "--iq0", "-", "48000", "16"], z.tobytes()).count(b"[OK]") >= 3
    _both("lms6Xmod", ["--lmsX", "-r", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], y.tobytes())        # forced 4797.8 Bd bit clock on the same samples
    # an LMS-X signal (300-byte blocks at 4797.8 Bd): with --lmsX the decoder sets dsp.br after init_buffers() and the seam follows (4720 raw bits per hit)
    w = synth.lms6_capture(sr=48_000, seconds=6.0, noise_sigma=0.05, seed=31, baud=4797.8, lmsx=True)
    out = _both("lms6Xmod", ["--lmsX", "-r", "--ecc", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], w.tobytes())
    assert out.count(b"[OK]") >= 5 and out.startswith(b"24 46 05 00")
    assert _both("lms6Xmod", ["--lmsX", "--vit", "--ecc", "--json", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], w.tobytes()).count(b'"id": "LMSX-') >= 5
    # without --lmsX the decoder recognises the type from the first block and changes dsp.br / dsp.sps in mid-stream (lms6Xmod.c:1436-1462): the seam sets
    # the engine up again from 64 bits before the end of that block (the reference carries its filter state over that point): the same decoded frames,
    # the raw bytes of frames with errors need not be the same
    args = ["--vit", "--ecc", "--json", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
    a = subprocess.run([os.path.join(REF, "lms6Xmod_seam")] + args, input=w.tobytes(), capture_output=True, timeout=300)
    b = subprocess.run([os.path.join(REF, "lms6Xmod")] + args, input=w.tobytes(), capture_output=True, timeout=300)
    assert a.returncode == b.returncode == 0, a.stderr[-400:]
    ja = [l for l in a.stdout.splitlines() if l.startswith(b"{")]
    jb = [l for l in b.stdout.splitlines() if l.startswith(b"{")]
    assert ja == jb and len(ja) >= 4 and b'"subtype": "LMSX-403"' in ja[0]
    ok = lambda o: [l for l in o.splitlines() if b"[OK]" in l]
    assert ok(a.stdout) == ok(b.stdout)


@pytest.mark.parametrize("sr", [250_000, 1_000_000, 1_200_000, 2_048_000, 3_200_000, 6_000_000])
def test_seam_rs41_sample_rates(sr):
    """SDR rates off the benchmark's 2.4 Msps: other decimation factors / tap counts (decM 5 .. 125: the runtime-D and the wide
    decimator variants), a designated IF above 48 kHz where 48000 does not divide the rate (2.048 Msps -> 51.2 kHz)."""
    from tools import synth
    fq = synth.snap_fq(-0.11, sr)
    x = synth.rs41_capture(sr=sr, seconds=3.3, fq=fq, noise_sigma=0.02, frame_kw=ECEF, n_frames=3, t_first=0.1, seed=97, bit_errors=2)
    out = _both("rs41mod", ["-r", "--ecc2", "--crc", "--IQ", repr(fq), "--lpIQ", "-", str(sr), "16"], x.tobytes())
    assert len(out.splitlines()) == 3 and out.count(b"[OK]") == 3


def test_generic_engine_batched_channels_python():
    """Engine(sonde="generic") from Python: three MTS01-style channels (1200 Bd, 32-symbol header) at different carriers in one engine give
    the same hits and soft bits as three single-channel engines — the batched form of what the seam does for one decoder process."""
    from tools import synth
    from radiosonde_auto_rx_amd.engine import Engine
    sr = 480_000
    f = synth.FAMILY["mts01mod"]
    gen = dict(header=f["header"], baud=f["baud"], bt=1.5, h=0.9, symlen=1, symhd=1, hdmax=2, bitofs=0, nbits=f["nbits"], l_win=2.0, lpiq_bw=4000, lpfm_bw=4000)
    fqs = [synth.snap_fq(v, sr) for v in (0.1, -0.23, 0.31)]
    caps = np.stack([synth.family_capture("mts01mod", sr=sr, seconds=3.2, fq=fq, seed=70 + k, t_first=0.2 + 0.1 * k) for k, fq in enumerate(fqs)])

    def run(fq_list, x):
        eng = Engine(fq_list, sr, sonde="generic", generic=gen, thres=0.76, max_chunk=sr, max_frames=16)
        hits = []
        n = x.shape[1] // 2
        for s0 in range(0, n, sr // 2):
            s1 = min(n, s0 + sr // 2)
            eng.process_host(np.ascontiguousarray(x[:, 2 * s0:2 * s1]))
            hits += eng.fetch_hits(finish=s1 >= n)
        eng.close()
        return hits

    allh = run(fqs, caps)
    for c, fq in enumerate(fqs):
        one = run([fq], caps[c:c + 1])
        mine = [h for h in allh if h["channel"] == c]
        assert len(one) == len(mine) >= 2
        for a, b in zip(mine, one):
            assert a["mv_pos"] == b["mv_pos"] and a["nbits"] == b["nbits"] == f["nbits"] and abs(a["mv"]) > 0.76
            assert np.array_equal(a["soft"], b["soft"])


@pytest.mark.parametrize("ecc", ["--ecc3", "--ecc4"])
def test_seam_rs41_ecc3_second_soft_bit(ecc):
    """--ecc3 / --ecc4 (rs41mod.c:2925, :1881-1950) slice every bit from the sum of read_softbit2p()'s two soft bits — the second one is the
    same window one IF sample earlier (cfg.keep_soft = 2, sonde_engine_fetch_soft1) — and use byte scores for erasure / bit-toggle
    decoding: at a noise level where that changes the correction counts, the reference's decoder on the seam prints what it prints on
    demod_mod.c."""
    from tools import synth
    outs = []
    for ns, seed in ((0.38, 55), (0.44, 56)):
        x = synth.rs41_capture(sr=48_000, seconds=8.3, fq=0.0, noise_sigma=ns, frame_kw=ECEF, n_frames=8, t_first=0.15, seed=seed)
        outs.append(_both("rs41mod", ["-r", ecc, "--crc", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], x.tobytes()))
        plain = subprocess.run([os.path.join(REF, "rs41mod"), "-r", "--ecc2", "--crc", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"], input=x.tobytes(),
                               capture_output=True, timeout=300).stdout
        assert plain != outs[-1]                                   # the second soft bit matters at this noise level
    assert outs[0].count(b"[OK]") >= 5


@pytest.mark.parametrize("ecc", ["--ecc3", "--ecc4"])
def test_native_rs41_ecc3_ecc4_match_reference(ecc):
    """host/bin/rs41mod --ecc3 / --ecc4: the list decoding (erasures from the byte scores, bit toggling, bytes known from earlier
    frames — rs41mod.c:1703-1974, :2490-2522, :2918-2962) runs in sonde_rs41_dec_ecc() over the library's own RS(255,231) codec
    (sonde_ecc.h), not in the reference's code: same stdout as the reference binary, raw and decoded, at noise levels where the
    plain --ecc2 output differs."""
    from tools import synth
    native = os.path.join(ROOT, "host", "bin", "rs41mod")
    ref = os.path.join(REF, "rs41mod")
    if not (os.path.exists(native) and os.path.exists(ref)):
        pytest.skip("host/bin or oracle/_ref not built")
    seam = os.path.join(REF, "rs41mod_seam")
    tail = ["--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    changed = 0
    for ns, seed in ((0.38, 55), (0.44, 56), (0.47, 57)):
        x = synth.rs41_capture(sr=48_000, seconds=12.3, fq=0.0, noise_sigma=ns, frame_kw=ECEF, n_frames=12, t_first=0.15, seed=seed).tobytes()
        for mode in (["-r", ecc, "--crc"], ["-v", ecc, "--crc", "--ptu"], ["--json", ecc]):
            a = subprocess.run([native] + mode + tail, input=x, capture_output=True, timeout=300, env=env)
            b = subprocess.run([ref] + mode + tail, input=x, capture_output=True, timeout=300)
            c = subprocess.run([seam] + mode + tail, input=x, capture_output=True, timeout=300)
            assert a.returncode == b.returncode == c.returncode == 0, a.stderr[-300:]
            # the reference's own rs41_ecc() on the very same soft bits (the seam): identical at every noise level
            assert a.stdout == c.stdout, (mode, ns, a.stdout[:400], c.stdout[:400])
            if ns < 0.46:
                assert a.stdout == b.stdout, (mode, ns, a.stdout[:400], b.stdout[:400])
            else:
                # deep in the noise an UNCORRECTABLE frame keeps the bits the last attempt toggled, and which bits those are depends on
                # the order of byte scores that differ by the float noise floor of the soft bits: decoded
                # frames must still be identical, failed ones may differ in a few characters
                la, lb = a.stdout.splitlines(), b.stdout.splitlines()
                assert len(la) == len(lb)
                for u, v in zip(la, lb):
                    if b"[OK]" in v or b"[OK]" in u:
                        assert u == v
                    else:
                        assert len(u) == len(v) and sum(p != q for p, q in zip(u, v)) <= 8
        plain = subprocess.run([ref, "-r", "--ecc2", "--crc"] + tail, input=x, capture_output=True, timeout=300).stdout
        changed += plain != subprocess.run([ref, "-r", ecc, "--crc"] + tail, input=x, capture_output=True, timeout=300).stdout
    assert changed >= 2


@pytest.mark.parametrize("binary,shift", [("rs41mod", "1"), ("rs41mod", "-2"), ("dfm09mod", "-1"), ("m10mod", "2")])
def test_seam_bit_offset_option(binary, shift):
    """-d <shift>: the decoders add it to the bitofs they pass to find_header() / read_softbit*(); the seam hands it to the engine
    (sonde_engine_set_sync) — soft bits move by whole IF samples, so correction counts / scores printed by the reference change with it."""
    from tools import synth
    if binary == "rs41mod":
        x = synth.rs41_capture(sr=48_000, seconds=4.3, fq=0.0, noise_sigma=0.3, frame_kw=ECEF, n_frames=4, t_first=0.15, seed=61)
        args = ["-r", "--ecc2", "--crc"]
    elif binary == "dfm09mod":
        x = synth.dfm_capture(sr=48_000, seconds=3.2, fq=0.0, noise_sigma=0.15, seed=62)
        args = ["-r", "--ecc2"]
    else:
        x = synth.m10_capture(sr=48_000, seconds=4.3, noise_sigma=0.1, seed=63, frame_fn=lambda k: synth.m10_frame(k, rng=np.random.default_rng(80 + k)))
        args = ["-r", "-v"]
    tail = ["--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
    out = _both(binary, args + ["-d", shift] + tail, x.tobytes())
    assert len(out.splitlines()) >= 2


def test_seam_rejects_what_it_cannot_reslice():
    """--spike (dfm09mod.c:1374, read_softbit(.., spike)): the reference's clipping reads an uninitialised local (demod_mod.c:1016,1046), so the
    seam ends the program with a message instead of silently returning unclipped bits; the same guard covers a per-call `l` / `ofs` other than
    what the hit was sliced with."""
    from tools import synth
    seam = os.path.join(REF, "dfm09mod_seam")
    if not os.path.exists(seam):
        pytest.skip("oracle/_ref seam binaries not built (make -C oracle ref)")
    x = synth.dfm_capture(sr=48_000, seconds=2.2, fq=0.0, noise_sigma=0.05, seed=5)
    wav = synth.wav_bytes(synth.fm_audio(x), 48_000)                      # spike only applies to FM audio / --iq0 input (dfm09mod.c:1693)
    r = subprocess.run([seam, "-r", "--ecc", "--spike"], input=wav, capture_output=True, timeout=300)
    assert r.returncode == 2 and b"spike" in r.stderr
    r = subprocess.run([seam, "-r", "--ecc"], input=wav, capture_output=True, timeout=300)
    assert r.returncode == 0 and len(r.stdout.splitlines()) >= 2
