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
    sc = Scanner(2_400_000, fq=[0.1], bw_khz=96.0)                      # wide IF: N_DFT = 16384 (the global-memory form of the exact kernel)
    assert sc.info["N"] == 16384 and sc.info["if_sr"] == 96000 and sc.info["decM"] == 25
    sc.close()
    with pytest.raises(SondeError):
        Scanner(1_000_000, n_channels=1, iq_mode=1)                     # an IF rate that would need N_DFT = 65536


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


def test_lagged_fetch_behind_a_long_running_decoder_sees_only_its_own_calls_frames():
    """Many channels, every frame damaged (the Reed-Solomon decoder on stream E takes its time), two streams, every fetch one call late, calls short enough that the
    next call's frame sync runs while the decoder of this one is still busy: what a call publishes is the frame counter as ITS last frame sync left it (taken on the
    frame sync's stream), not the live counter behind the decoder — which may already include slots the next call has counted but not written.
    The lagged run must deliver the frames of the plain run: same records, nothing stale, nothing skipped."""
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    sr = 480_000
    C = 48
    fqs = [synth.snap_fq(0.004 * (k - C / 2), sr) for k in range(C)]
    caps = [synth.rs41_capture(sr=sr, seconds=2.4, fq=fq, seed=300 + k, noise_sigma=0.03, bit_errors=18 + (k % 7), t_first=0.05 + 0.013 * (k % 9)) for k, fq in enumerate(fqs)]
    n = min(len(c) for c in caps) // 2 // 10 * 10
    xb = np.stack([c[:2 * n] for c in caps])
    out = {}
    for pipe in (False, True):
        eng = Engine(fqs, sr, ecc=2, max_chunk=48_000, keep_soft=False, pipeline=pipe, max_frames=8 * C)
        frames = []
        for pos in range(0, n, 48_000):
            take = min(48_000, n - pos)
            eng.process_host(np.ascontiguousarray(xb[:, 2 * pos:2 * (pos + take)]))
            fr = eng.fetch_frames_np(lag=1 if pipe else 0)
            frames += [(int(f["channel"]), int(f["mv_pos"]), int(f["ecc"]), bytes(f["frame"])) for f in fr]
        fr = eng.fetch_frames_np(lag=0)
        frames += [(int(f["channel"]), int(f["mv_pos"]), int(f["ecc"]), bytes(f["frame"])) for f in fr]
        assert eng.host_ecc_frames() == 0
        out[pipe] = sorted(frames)
        eng.close()
    assert len(out[False]) >= C and out[True] == out[False]
    assert sum(1 for f in out[False] if f[2] > 0) >= C // 2                 # the decoder had work


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


@pytest.mark.gpu
def test_base_rate_channel_restart_behaves_like_a_fresh_engine():
    """The same for a 2.4 Msps `--IQ fq` engine (mixer + decimator in front of the IF-rate chain; the hand-scheduled decimator): channel 1
    carries a first sonde, is ended, gets ANOTHER carrier (sonde_engine_tune_channel) and is restarted in the middle of the engine's life — the
    mixer table phase, the IQ-DC mean with its segment schedule and the decimator history of that channel start over, calls are cut at every
    channel's own segment edges from then on.  The second stream decodes to exactly what a fresh one-channel engine gives (frame bytes, ECC
    verdict, header positions counted from the stream's own start, soft bits within 1e-5), channel 0 carries on undisturbed."""
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    sr, blk = 2_400_000, 240_000
    fqs = [synth.snap_fq(f, sr) for f in (0.11, -0.23, 0.31)]
    caps = [synth.rs41_capture(sr=sr, seconds=5.4, fq=fqs[0], noise_sigma=0.03, n_frames=5, t_first=0.2, seed=171, dc=0.01 - 0.02j),
            synth.rs41_capture(sr=sr, seconds=2.3, fq=fqs[1], noise_sigma=0.05, n_frames=2, t_first=0.31, seed=172, first_frame_no=77),
            synth.rs41_capture(sr=sr, seconds=2.7, fq=fqs[2], noise_sigma=0.04, n_frames=2, t_first=0.12, seed=173, first_frame_no=990, dc=-0.02 + 0.01j)]

    def alone(x, fq):
        e = Engine([fq], sr, max_chunk=sr, keep_soft=True)
        n = len(x) // 2 // 50 * 50
        out = []
        for s0 in range(0, n, blk):
            e.process_host(x[2 * s0:2 * min(n, s0 + blk)][None, :])
            out += e.fetch_frames(with_soft=True, finish=(s0 + blk >= n))
        e.close()
        return out

    want = [alone(c, f) for c, f in zip(caps, fqs)]
    assert [len(w) for w in want] == [5, 2, 2]
    eng = Engine([fqs[0], fqs[1]], sr, max_chunk=sr, keep_soft=True)
    n0 = len(caps[0]) // 2 // 50 * 50
    n1 = len(caps[1]) // 2 // 50 * 50
    n2 = len(caps[2]) // 2 // 50 * 50
    got0, got1, got2 = [], [], []
    seg, pos1 = 1, 0
    for s0 in range(0, n0, blk):
        a = caps[0][2 * s0:2 * min(n0, s0 + blk)]
        take = len(a) // 2
        src, nsrc = (caps[1], n1) if seg == 1 else (caps[2], n2)
        b = src[2 * pos1:2 * min(nsrc, pos1 + take)]
        if len(b) < len(a):
            b = np.concatenate([b, np.zeros(len(a) - len(b), np.int16)])
        pos1 += take
        eng.process_host(np.stack([a, b]))
        for f in eng.fetch_frames(with_soft=True):
            (got0 if f["channel"] == 0 else got1 if seg == 1 else got2).append(f)
        if seg == 1 and pos1 >= n1:                           # end of the first sonde on channel 1: its frame in progress, then the new carrier
            eng.finish_channel(1)
            for f in eng.fetch_frames(with_soft=True):
                (got0 if f["channel"] == 0 else got1).append(f)
            eng.tune_channel(1, fqs[2])
            eng.restart_channel(1)
            seg, pos1 = 2, 0
    for f in eng.fetch_frames(with_soft=True, finish=True):
        (got0 if f["channel"] == 0 else got2).append(f)
    eng.close()

    def same(a, b):
        assert len(a) == len(b), (len(a), len(b))
        for u, v in zip(a, b):
            assert u["mv_pos"] == v["mv_pos"] and u["ecc"] == v["ecc"] and u["len"] == v["len"] and bytes(u["frame"]) == bytes(v["frame"])
            nb = (u["nbytes"] - 8) * 8
            assert float(np.sqrt(np.mean((u["soft"][:nb] - v["soft"][:nb]) ** 2))) < 1e-5
    same(got0, want[0])
    same(got2, want[2][:len(got2)])
    assert len(got2) >= 1 and got2[0]["mv_pos"] == want[2][0]["mv_pos"]
    assert len(got1) >= 1 and bytes(got1[0]["frame"]) == bytes(want[1][0]["frame"]) and got1[0]["mv_pos"] == want[1][0]["mv_pos"]


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["audio", "iq2"])
def test_channel_restart_behaves_like_a_fresh_engine(form):
    """sonde_engine_finish_channel / sonde_engine_restart_channel (what the resident broker needs for decoder processes that come and go):
    a channel that has carried one stream is ended — its frame in progress comes out with the bits that exist — and restarted in the middle of
    the engine's life; the second stream decodes to exactly what a fresh one-channel engine gives (frame bytes, ECC, header positions counted
    from the stream's own start), while the neighbouring channel carries on undisturbed."""
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    ecef = dict(ecef_cm=(418833319, 85974133, 473346430))
    sr = 48_000
    caps = [synth.rs41_capture(sr=sr, seconds=5.4, fq=0.0, noise_sigma=0.03, frame_kw=ecef, n_frames=5, t_first=0.2, seed=71),
            synth.rs41_capture(sr=sr, seconds=3.1, fq=0.0, noise_sigma=0.05, frame_kw=ecef, n_frames=3, t_first=0.31, seed=72, first_frame_no=77),
            synth.rs41_capture(sr=sr, seconds=2.6, fq=0.0, noise_sigma=0.04, frame_kw=ecef, n_frames=2, t_first=0.12, seed=73, first_frame_no=990)]
    if form == "audio":
        caps = [np.ascontiguousarray(synth.fm_audio(c)) for c in caps]
        kw = dict(audio=True, lp_iq=False); per = 1
    else:
        kw = dict(iq_mode=2, lp_iq=True); per = 2

    def alone(x):
        e = Engine([0.0], sr, max_chunk=sr, **kw)
        n = len(x) // per
        out = []
        for s0 in range(0, n, 4800):
            e.process_host(x[per * s0:per * min(n, s0 + 4800)][None, :])
            out += e.fetch_frames(finish=(s0 + 4800 >= n))
        e.close()
        return out

    want = [alone(c) for c in caps]
    assert all(len(w) >= 2 for w in want)
    eng = Engine([0.0, 0.0], sr, max_chunk=sr, **kw)
    blk = 4800
    got = {0: [], 1: []}
    # channel 0: stream 0 all the way; channel 1: stream 1 from the start, ended early, then stream 2 after a restart
    plan1 = [(caps[1], len(caps[1]) // per - 9000), (caps[2], len(caps[2]) // per)]        # (stream, samples to feed): stream 1 is cut inside its last frame
    seg, pos1 = 0, 0
    n0 = len(caps[0]) // per
    sil = np.zeros(per * blk, np.int16)
    out_seg = {0: [], 1: []}
    for s0 in range(0, n0, blk):
        a = caps[0][per * s0:per * (s0 + blk)]
        if len(a) < per * blk:
            a = np.concatenate([a, sil[:per * blk - len(a)]])
        x1, n1 = plan1[seg] if seg < len(plan1) else (None, 0)
        b = sil
        if x1 is not None:
            b = x1[per * pos1:per * min(n1, pos1 + blk)]
            if len(b) < per * blk:                        # the stream ends inside this block: silence behind it until the restart
                b = np.concatenate([b, sil[:per * blk - len(b)]])
            pos1 += blk
        eng.process_host(np.stack([a, b]))
        for f in eng.fetch_frames():
            (got[0] if f["channel"] == 0 else out_seg[seg if seg < len(plan1) else len(plan1) - 1]).append(f)
        if x1 is not None and pos1 >= n1:                 # end of this stream on channel 1
            eng.finish_channel(1)
            for f in eng.fetch_frames():
                (got[0] if f["channel"] == 0 else out_seg[seg]).append(f)
            eng.restart_channel(1)
            seg += 1; pos1 = 0
    for f in eng.fetch_frames(finish=True):
        (got[0] if f["channel"] == 0 else out_seg[min(seg, len(plan1) - 1)]).append(f)
    eng.close()

    def same(a, b, full=True):
        assert len(a) == len(b), (len(a), len(b))
        for u, v in zip(a, b):
            assert u["mv_pos"] == v["mv_pos"] and u["ecc"] == v["ecc"] and u["len"] == v["len"]
            assert bytes(u["frame"]) == bytes(v["frame"])
    same(got[0], want[0])
    same(out_seg[1], want[2])                             # the restarted channel: like a fresh engine
    assert len(out_seg[0]) >= 2                           # the first stream on channel 1 up to its cut: the complete frames are the stand-alone ones
    for u, v in zip(out_seg[0][:-1], want[1]):
        assert u["mv_pos"] == v["mv_pos"] and bytes(u["frame"]) == bytes(v["frame"])


@pytest.mark.gpu
def test_fetching_again_behind_an_ended_channel_returns_nothing():
    """A fetch, sonde_engine_finish_channel, a fetch that takes the frame in progress, and ANOTHER fetch with no process call in between (the
    broker drains a channel's queue until a fetch comes back empty): the last one returns no frame and reports no overflow — the frame counter
    snapshot of the last process call is older than what the end-of-stream frame sync added, and must not pull the read index back."""
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    sr = 48_000
    x = synth.rs41_capture(sr=sr, seconds=2.0, fq=0.0, noise_sigma=0.03, n_frames=2, t_first=0.2, seed=81)       # the second frame is cut by the end
    eng = Engine([0.0, 0.0], sr, max_chunk=sr, iq_mode=2, lp_iq=True, max_frames=8)
    n = len(x) // 2
    frames = []
    for s0 in range(0, n, 9600):
        blk = x[2 * s0:2 * min(n, s0 + 9600)]
        eng.process_host(np.stack([blk, np.zeros_like(blk)]))
        frames += eng.fetch_frames()
    assert len(frames) == 1
    eng.finish_channel(0)
    tail = eng.fetch_frames()
    assert len(tail) == 1 and tail[0]["channel"] == 0 and tail[0]["mv_pos"] > frames[0]["mv_pos"]
    assert eng.fetch_frames() == [] and eng.fetch_frames() == []
    assert eng.overflowed() is False
    eng.close()


def test_slot_sized_sync_kernels_give_identical_frames(monkeypatch):
    """SONDE_SMALL_TAIL=1: the header search's window transform in half the LDS (two 4096-point halves, the waiting half parked in global memory) and the
    256-thread frame sync — same butterflies on the same operands: header scores, positions, soft bits and frames identical to the bit"""
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    sr = 480_000
    fqs = [synth.snap_fq(f, sr) for f in (-0.21, 0.07, 0.33)]
    caps = [synth.rs41_capture(sr=sr, seconds=3.3, fq=fq, seed=70 + k, noise_sigma=0.05 + 0.1 * k, bit_errors=5 * k, t_first=0.1 + 0.23 * k) for k, fq in enumerate(fqs)]
    n = min(len(c) for c in caps) // 2 // 10 * 10
    x = np.stack([c[:2 * n] for c in caps])

    def run():
        eng = Engine(fqs, sr, ecc=2, max_chunk=sr, keep_soft=True)
        out = []
        for pos in range(0, n, sr):
            take = min(sr, n - pos) // 10 * 10
            eng.process_host(np.ascontiguousarray(x[:, 2 * pos:2 * (pos + take)]))
            out += eng.fetch_frames(with_soft=True)
        out += eng.fetch_frames(with_soft=True, finish=True)
        eng.close()
        return sorted(out, key=lambda f: (f["channel"], f["mv_pos"]))

    a = run()
    monkeypatch.setenv("SONDE_SMALL_TAIL", "1")
    b = run()
    assert len(a) == len(b) >= 9
    for fa, fb in zip(a, b):
        assert (fa["channel"], fa["mv_pos"], fa["line"], fa["ecc"]) == (fb["channel"], fb["mv_pos"], fb["line"], fb["ecc"])
        assert fa["mv"] == fb["mv"] and np.array_equal(fa["soft"], fb["soft"])


@pytest.mark.parametrize("bits", [16, 8])
def test_iq_dec_with_a_decimator_of_more_than_eight_tap_columns_matches_the_reference(bits):
    """`iq_dec --IFbw 32` (the narrowest IF the reference accepts, iq_dec.c:993-1000) at 960 kHz: 321 taps at a decimation of 30 are 11 tap columns — more than the
    packed decimator kernels hold.  Such configurations run through the plain float32 mixer / FIR kernels behind a conversion of the 16- / 8-bit samples
    (found missing by tests/fuzz/fuzz_iqdec.py: the CLI used to end with 255).  IQ and FM output against the compiled reference on the same bytes."""
    from tools import synth
    ref = os.path.join(ROOT, "oracle", "_ref", "iq_dec")
    if not os.path.exists(ref):
        pytest.fail("oracle/_ref/iq_dec missing: run __graft_entry__.build() where /root/reference exists")
    sr = 960_000
    fq = synth.snap_fq(-0.21, sr)
    x = synth.rs41_capture(sr=sr, seconds=1.2, fq=fq, seed=4, noise_sigma=0.03, t_first=0.05)
    data = (x if bits == 16 else synth.to_u8(x)).tobytes()
    for opts, dt, tol in ((["--bo", "32", "--iq", repr(fq), "--IFbw", "32"], np.float32, 2e-6), (["--bo", "16", "--iq", repr(fq), "--IFbw", "32", "--FM", "--lpFM"], np.int16, 0.5)):
        a = subprocess.run([os.path.join(BIN, "iq_dec")] + opts + ["-", str(sr), str(bits)], input=data, capture_output=True, timeout=120)
        b = subprocess.run([ref] + opts + ["-", str(sr), str(bits)], input=data, capture_output=True, timeout=120)
        assert a.returncode == b.returncode == 0 and len(a.stdout) == len(b.stdout) > 0, (opts, a.returncode, a.stderr[-200:])
        assert a.stderr.decode().splitlines()[:2] == b.stderr.decode().splitlines()[:2] == ["IF: 32000", "dec: 30"]
        pa, pb = np.frombuffer(a.stdout, dt).astype(np.float64), np.frombuffer(b.stdout, dt).astype(np.float64)
        assert np.sqrt(np.mean(np.square(pa - pb))) <= tol, (opts, float(np.sqrt(np.mean(np.square(pa - pb)))))
