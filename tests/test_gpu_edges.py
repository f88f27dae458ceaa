"""Edge cases through the C ABI / CLIs on the GPU: empty and sub-block inputs, one-block chunks (shorter than the
decimator's history), engine limits, and error returns instead of silent approximations."""
import os
import subprocess

import numpy as np
import pytest
from golden_cases import capture, load

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "bin")


def _run(cmd, data=b""):
    return subprocess.run(cmd, input=data, capture_output=True, timeout=60)


def test_cli_empty_and_short_inputs():
    """EOF before the first block: the reference tools print their header lines and exit 0 with no output."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    short = np.zeros(2 * 17, np.int16).tobytes()                      # 17 samples < one decimation block of 50
    for data in (b"", short):
        r = _run([os.path.join(BIN, "rs41mod"), "-r", "--ecc2", "--IQ", "0.1", "--lpIQ", "-", "2400000", "16"], data)
        assert r.returncode == 0 and r.stdout == b"" and r.stderr.decode().splitlines()[:2] == ["IF: 48000", "dec: 50"]
        r = _run([os.path.join(BIN, "dfm09mod"), "-r", "--ecc", "--IQ", "0.1", "--lpIQ", "-", "2400000", "16"], data)
        assert r.returncode == 0 and r.stdout == b""
        r = _run([os.path.join(BIN, "dft_detect"), "--IQ", "0.1", "--dc", "-", "2400000", "16"], data)
        assert r.returncode == 0 and r.stdout == b""
        r = _run([os.path.join(BIN, "iq_dec"), "--bo", "16", "-", "2400000", "16"], data)
        assert r.returncode == 0 and r.stdout == b""
        r = _run([os.path.join(BIN, "fsk_demod"), "--cs16", "-s", "2", "48000", "4800", "-", "-"], data)
        assert r.returncode == 0 and r.stdout == b""
    # argument errors: 255 like the reference's `return -1`
    r = _run([os.path.join(BIN, "rs41mod"), "-r", "--IQ", "0.1", "-", "0", "16"])
    assert r.returncode == 255
    r = _run([os.path.join(BIN, "rs41mod"), "-r", "--ecc5", "--IQ", "0.1", "-", "2400000", "16"])
    assert r.returncode == 255 and b"not supported" in r.stderr          # unknown options: refused, not ignored
    r = _run([os.path.join(BIN, "rs41mod"), "-r", "--ecc3", "--rawhex", "-"])
    assert r.returncode == 255                                           # --ecc3/4 work on the demodulator's soft bits only
    r = _run([os.path.join(BIN, "rs41mod"), "-r", "--noLUT", "--dc", "--IQ", "0.1", "-", "2400000", "16"])
    assert r.returncode == 255                                           # --noLUT with --dc (Df inside the base-rate mixer): refused


def test_one_block_chunks_keep_the_decimator_history():
    """Process calls of a single decimation block (shorter than the Q-1 blocks of FIR history the P tail carries):
    same frames and same IF stream as one-second calls."""
    from radiosonde_auto_rx_amd.engine import Engine, TAP_IFIQ
    x, fq, sr = capture("rs41_480k_clean")
    g = load("rs41_480k_clean")
    D = 10
    n = len(x) // 2
    eng = Engine([fq], sr, keep_soft=True, max_chunk=sr)
    pos, lines = 0, []
    while pos < 4000 * D:                                             # 4000 calls of one block each ...
        eng.process_host(x[2 * pos:2 * (pos + D)]); pos += D
    a = eng.read_tap(0, TAP_IFIQ, 0, 4000)
    while pos < n - n % D:                                            # ... then the rest in big calls
        take = min(sr, n - n % D - pos)
        eng.process_host(x[2 * pos:2 * (pos + take)]); pos += take
        lines += [f["line"] for f in eng.fetch_frames()]
    lines += [f["line"] for f in eng.fetch_frames(finish=True)]
    eng.close()
    ref = Engine([fq], sr, keep_soft=True, max_chunk=sr)
    ref.process_host(x[:2 * sr])
    b = ref.read_tap(0, TAP_IFIQ, 0, 4000)
    ref.close()
    assert np.array_equal(a, b)
    assert lines == g["lines"]


def test_engine_rejects_bad_requests():
    from radiosonde_auto_rx_amd.engine import Engine, SondeError
    eng = Engine([0.1], 2_400_000, max_chunk=240_000)
    with pytest.raises(SondeError):
        eng.process_host(np.zeros(2 * 240_050, np.int16))               # larger than max_chunk
    with pytest.raises(SondeError):
        eng.process_host(np.zeros(2 * 1234, np.int16))                  # not a multiple of decM
    assert eng.fetch_frames() == []
    eng.close()
    with pytest.raises(SondeError):
        Engine([0.1], 2_400_000, bits=24)                               # only 8 / 16 / 32-bit samples exist
    from radiosonde_auto_rx_amd.fsk import FskModem
    with pytest.raises(SondeError):
        FskModem(48000, 4799)                                           # Fs % Rs != 0 (the reference asserts)
    from radiosonde_auto_rx_amd.scan import Scanner
    with pytest.raises(SondeError):
        Scanner(2_400_000, fq=[0.1], bw_khz=96.0)                       # wide IF needs N_DFT > 8192: refused


def test_two_stream_pipeline_matches_single_stream():
    """pipeline=1 (IF-rate kernels on a second stream, frames fetched one call late) gives the frames of the plain engine: the
    decimator of call k+1 may overlap the IF-rate kernels of call k, but must not run two calls ahead of them (ring reuse)."""
    from radiosonde_auto_rx_amd.engine import Engine
    x, fq, sr = capture("rs41_480k_be30")
    n = len(x) // 2
    D = 10
    out = {}
    for pipe in (False, True):
        eng = Engine([fq] * 3, sr, max_chunk=48_000, keep_soft=False, pipeline=pipe, max_frames=64)
        xb = np.stack([x, x, x])
        frames = []
        for pos in range(0, n - n % D, 48_000):                       # 0.1 s calls: the streams really overlap
            take = min(48_000, n - n % D - pos)
            eng.process_host(np.ascontiguousarray(xb[:, 2 * pos:2 * (pos + take)]))
            fr = eng.fetch_frames_np(lag=1 if pipe else 0)
            frames += [(int(f["channel"]), int(f["mv_pos"]), bytes(f["frame"])) for f in fr]
        fr = eng.fetch_frames_np(lag=0)
        frames += [(int(f["channel"]), int(f["mv_pos"]), bytes(f["frame"])) for f in fr]
        out[pipe] = sorted(frames)
        eng.close()
    assert out[True] == out[False] and len(out[False]) >= 3
    with pytest.raises(Exception):                                      # FM audio writes the rings of stream B on stream A: refused
        Engine([0.0], 48000, audio=True, pipeline=True)


def test_long_chunk_is_fully_consumed():
    """A call much longer than 64 correlation windows (K - 4 = 7508 IF samples each): the frame sync keeps going until the samples
    are used up — all frames of a 12 s capture come out of ONE call."""
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    sr = 480_000
    fq = synth.snap_fq(0.07, sr)
    x = synth.rs41_capture(sr=sr, seconds=12.3, fq=fq, seed=5, noise_sigma=0.02)
    n = (len(x) // 2 // 10) * 10
    ref = Engine([fq], sr, max_chunk=sr, max_frames=32)
    want = []
    for pos in range(0, n, sr):
        take = min(sr, n - pos)
        ref.process_host(x[2 * pos:2 * (pos + take)])
        want += [f["line"] for f in ref.fetch_frames()]
    ref.close()
    eng = Engine([fq], sr, max_chunk=n, max_frames=32)
    eng.process_host(x[:2 * n])
    got = [f["line"] for f in eng.fetch_frames()]
    eng.close()
    assert len(want) >= 11 and got == want


def test_queue_overflow_is_reported_not_returned_as_error():
    """max_frames smaller than what one call produces: the fetch returns the frames that survived, overflowed() says that older ones
    were overwritten (and clears)."""
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    sr = 480_000
    fq = synth.snap_fq(-0.11, sr)
    x = synth.rs41_capture(sr=sr, seconds=5.3, fq=fq, seed=9, noise_sigma=0.02)
    n = (len(x) // 2 // 10) * 10
    eng = Engine([fq], sr, max_chunk=n, max_frames=2)
    eng.process_host(x[:2 * n])
    fr = eng.fetch_frames()
    assert len(fr) == 2 and eng.overflowed() is True and eng.overflowed() is False
    eng.close()
