"""The mixed-type engine (sonde_engine_create_mixed, include/sonde_hip.h): RS41, DFM09 and M10 channels side by side behind ONE decimator launch per call
(nothing in front of the IF rate depends on the sonde type, /root/reference demod_mod.c:1222-1249), the IF-rate stages per type.

What must hold: every channel's frames are what a single-type engine — and therefore the reference decoder (test_gpu_parity.py, test_gpu_m10.py, and here
again against the compiled reference's stdout) — delivers on the same samples: text lines, header positions and scores bit for bit; for any order of the
types among the channels, any chunking (IQ-DC segment edges inside a call), lagged fetches, 8-bit input, a sample rate off the hand-scheduled decimator."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REF = {"rs41": ("rs41mod", ["-r", "--ecc2"]), "dfm": ("dfm09mod", ["-r", "--ecc"]), "m10": ("m10mod", ["-r", "-v"])}


def _capture(kind, k, sr, seconds, fq):
    from tools import synth
    if kind == "rs41":
        return synth.rs41_capture(sr=sr, seconds=seconds, fq=fq, seed=900 + k, noise_sigma=0.02 + 0.05 * (k % 3), bit_errors=(0, 6, 14, 30)[k % 4], t_first=0.12 + 0.05 * k)
    if kind == "dfm":
        return synth.dfm_capture(sr=sr, seconds=seconds, fq=fq, noise_sigma=0.02 + 0.1 * (k % 3), seed=910 + k, bit_errors_per_frame=k % 3, t_first=0.02 + 0.03 * k)
    return synth.m10_capture(sr=sr, seconds=seconds, fq=fq, noise_sigma=0.02 + 0.08 * (k % 3), seed=920 + k, t_first=0.2 + 0.07 * k,
                             frame_fn=lambda i, k=k: synth.m10_frame(i, rng=np.random.default_rng(700 + 10 * k + i), good_checksum=(i + k) % 4 != 3))


def _bank(kinds, sr, seconds):
    from tools import synth
    rng = np.random.default_rng(77)
    fqs = [synth.snap_fq(float(rng.uniform(-0.4, 0.4)), sr) for _ in kinds]
    x = np.stack([_capture(kd, k, sr, seconds, fqs[k]) for k, kd in enumerate(kinds)])
    return fqs, x


def _fetch_all(eng, kinds_present, finish, lag=0):
    out = []
    if "rs41" in kinds_present:
        if lag and not finish:
            fr = eng.fetch_frames_np(lag=lag)
            from radiosonde_auto_rx_amd.engine import lib, SondeFrame
            import ctypes as C
            line = C.create_string_buffer(1200)
            for f in fr:
                sf = SondeFrame.from_buffer_copy(f.tobytes())
                ll = lib().sonde_rs41_rawline(C.byref(sf), line, 1200)
                out.append(dict(kind="rs41", channel=int(f["channel"]), mv=float(f["mv"]), mv_pos=int(f["mv_pos"]), line=line.raw[:ll].decode()))
        else:
            out += [dict(kind="rs41", channel=f["channel"], mv=f["mv"], mv_pos=f["mv_pos"], line=f["line"]) for f in eng.fetch_frames(finish=finish)]
    if "dfm" in kinds_present:
        if lag and not finish:
            import ctypes as C
            from radiosonde_auto_rx_amd.engine import lib
            buf, n = eng.fetch_dfm_raw(lag=lag)
            line = C.create_string_buffer(128)
            for i in range(n):
                ll = lib().sonde_dfm_rawline(C.byref(buf[i]), eng.ecc, line, 128)
                out.append(dict(kind="dfm", channel=buf[i].channel, mv=buf[i].mv, mv_pos=buf[i].mv_pos, line=line.raw[:ll].decode()))
        else:
            out += [dict(kind="dfm", channel=f["channel"], mv=f["mv"], mv_pos=f["mv_pos"], line=f["line"]) for f in eng.fetch_dfm(finish=finish)]
    if "m10" in kinds_present:
        if lag and not finish:
            import ctypes as C
            from radiosonde_auto_rx_amd.engine import lib
            buf, n = eng.fetch_m10_raw(lag=lag)
            line = C.create_string_buffer(420)
            for i in range(n):
                ll = lib().sonde_m10_rawline(C.byref(buf[i]), 1, line, 420)
                out.append(dict(kind="m10", channel=buf[i].channel, mv=buf[i].mv, mv_pos=buf[i].mv_pos, line=line.raw[:max(ll, 0)].decode()))
        else:
            out += [dict(kind="m10", channel=f["channel"], mv=f["mv"], mv_pos=f["mv_pos"], line=f["line"]) for f in eng.fetch_mxx(finish=finish)]
    return out


def _run_mixed(fqs, kinds, x, sr, chunk, lag=0, bits=16, **kw):
    from radiosonde_auto_rx_amd.engine import MixedEngine
    eng = MixedEngine(fqs, kinds, sr, max_chunk=max(chunk, sr), max_frames=16 * len(kinds), bits=bits, **kw)
    D = eng.info["decM"]
    n = x.shape[-1] // 2
    got = []
    present = set(kinds)
    pos = 0
    while pos < n:
        take = min(chunk, n - pos) // D * D
        if take <= 0:
            break
        eng.process_host(x[..., 2 * pos:2 * (pos + take)])
        pos += take
        got += _fetch_all(eng, present, False, lag=lag)
    got += _fetch_all(eng, present, True)
    assert not eng.overflowed()
    eng.close()
    return got


def _run_single(fqs, kinds, x, sr, chunk, bits=16):
    """every type on an engine of its own — what the mixed engine must reproduce; channel numbers translated back"""
    from radiosonde_auto_rx_amd.engine import Engine
    got = []
    for kind in dict.fromkeys(kinds):
        idx = [c for c, kd in enumerate(kinds) if kd == kind]
        eng = Engine([fqs[c] for c in idx], sr, sonde=kind, ecc={"rs41": 2, "dfm": 1, "m10": 0}[kind], max_chunk=max(chunk, sr), max_frames=16 * len(idx), bits=bits)
        D = eng.info["decM"]
        n = x.shape[-1] // 2
        xs = np.ascontiguousarray(x[idx])
        pos = 0
        while pos < n:
            take = min(chunk, n - pos) // D * D
            if take <= 0:
                break
            eng.process_host(xs[..., 2 * pos:2 * (pos + take)])
            pos += take
            fr = _fetch_all(eng, {kind}, False)
            for f in fr:
                f["channel"] = idx[f["channel"]]
            got += fr
        fr = _fetch_all(eng, {kind}, True)
        for f in fr:
            f["channel"] = idx[f["channel"]]
        got += fr
        eng.close()
    return got


def _per_channel(frames, scores=True):
    d = {}
    for f in frames:
        d.setdefault(f["channel"], []).append((f["kind"], f["line"].rstrip(), f["mv_pos"]) + ((np.float32(f["mv"]).tobytes(),) if scores else ()))
    return d


def _scores(frames):
    d = {}
    for f in frames:
        d.setdefault(f["channel"], []).append(float(f["mv"]))
    return d


KINDS = ["rs41", "dfm", "m10", "m10", "rs41", "dfm", "rs41", "dfm", "m10", "rs41", "rs41"]


@pytest.fixture(scope="module")
def bank24():
    sr = 2_400_000
    fqs, x = _bank(KINDS, sr, 2.2)
    return sr, fqs, x


def test_mixed_engine_frames_equal_three_single_type_engines(bank24):
    sr, fqs, x = bank24
    a = _per_channel(_run_mixed(fqs, KINDS, x, sr, sr))
    b = _per_channel(_run_single(fqs, KINDS, x, sr, sr))
    assert sorted(a) == list(range(len(KINDS)))                     # every channel decoded something
    for c, kd in enumerate(KINDS):
        assert all(f[0] == kd for f in a[c])                        # ... of its own type, under the caller's channel number
    assert a == b
    assert sum(len(v) for v in a.values()) >= 30


def test_mixed_engine_frames_equal_the_reference_decoders(bank24):
    from oracle import bind
    if not bind.have_ref():
        pytest.fail("oracle/_ref (the compiled reference) is missing: run __graft_entry__.build() where /root/reference exists")
    sr, fqs, x = bank24
    a = _per_channel(_run_mixed(fqs, KINDS, x, sr, sr))
    for c, kd in enumerate(KINDS):
        exe, args = REF[kd]
        r = subprocess.run([os.path.join(bind.REFDIR, exe)] + args + ["--IQ", repr(fqs[c]), "--lpIQ", "-", str(sr), "16"], input=x[c].tobytes(), capture_output=True, timeout=300)
        want = [ln.rstrip() for ln in r.stdout.decode().splitlines()]
        assert len(want) >= 1
        assert [f[1] for f in a[c]] == want, (c, kd)


@pytest.mark.parametrize("chunk", [937_500, 1_000_050])
def test_mixed_engine_any_chunking_and_lagged_fetch(bank24, chunk):
    """calls that straddle the IQ-DC segment edges (75000 * 2^k samples, demod_mod.c:495-504: several decimator launches per call) and fetches one call behind"""
    sr, fqs, x = bank24
    r0, r1 = _run_mixed(fqs, KINDS, x, sr, sr), _run_mixed(fqs, KINDS, x, sr, chunk, lag=1)
    assert _per_channel(r1, scores=False) == _per_channel(r0, scores=False)
    # (where a segment edge falls inside a launch the first Q - 1 outputs behind it are corrected after the fact — md_dc_boundary, DESIGN §4.1: 2e-8 on the
    # IF stream, which can move a header score by a float ulp; lines and positions are exact)
    s0, s1 = _scores(r0), _scores(r1)
    for c in s0:
        assert np.allclose(s0[c], s1[c], rtol=0, atol=1e-6)


def test_mixed_engine_order_of_types_does_not_matter(bank24):
    sr, fqs, x = bank24
    ref = _per_channel(_run_mixed(fqs, KINDS, x, sr, sr))
    perm = [10, 3, 7, 0, 5, 1, 9, 2, 8, 6, 4]
    b = _per_channel(_run_mixed([fqs[p] for p in perm], [KINDS[p] for p in perm], np.ascontiguousarray(x[perm]), sr, sr))
    assert {perm[c]: v for c, v in b.items()} == ref


def test_mixed_engine_summaries_carry_the_callers_channel_numbers(bank24):
    import torch
    from radiosonde_auto_rx_amd.engine import MixedEngine
    from radiosonde_auto_rx_amd import shard
    sr, fqs, x = bank24
    eng = MixedEngine(fqs, KINDS, sr, max_chunk=sr, max_frames=16 * len(KINDS))
    buf = shard.summary_buffer(len(KINDS), torch.device("cuda:0"))
    eng.set_summary(buf.data_ptr(), 1000)
    for s in range(2):
        eng.process_host(x[:, 2 * s * sr:2 * (s + 1) * sr])
    eng.process_host(x[:, 4 * sr:])                                 # (0.2 s: the DFM hits that began late in the first second are complete now)
    eng.sync()
    torch.cuda.synchronize()
    rec = shard.decode_summaries(buf)
    counts = {}
    for f in _fetch_all(eng, set(KINDS), False):
        counts[f["channel"]] = counts.get(f["channel"], 0) + 1
    tcode = {"rs41": 41, "dfm": 9, "m10": 10}
    for c, kd in enumerate(KINDS):
        assert int(rec["channel_id"][c]) == 1000 + c and int(rec["type"][c]) == tcode[kd]
        hits = counts.get(c, 0) if kd != "dfm" else (counts.get(c, 0) + 7) // 8          # (a DFM hit = up to eight frames)
        assert int(rec["frames"][c]) == hits, (c, kd)
    eng.close()


def test_mixed_engine_8bit_input_and_generic_decimator():
    """sr = 480 kHz (D = 10: the templated decimator, channel rows through in_row there too), cu8 samples through k_u8_to_s16"""
    sr = 480_000
    kinds = ["dfm", "rs41", "m10", "rs41", "dfm"]
    fqs, x = _bank(kinds, sr, 2.3)
    a = _per_channel(_run_mixed(fqs, kinds, x, sr, sr))
    b = _per_channel(_run_single(fqs, kinds, x, sr, sr))
    assert a == b and sorted(a) == list(range(len(kinds)))
    x8 = ((x.astype(np.int32) >> 8) + 128).astype(np.uint8)
    a8 = _per_channel(_run_mixed(fqs, kinds, x8, sr, sr, bits=8))
    b8 = _per_channel(_run_single(fqs, kinds, x8, sr, sr, bits=8))
    assert a8 == b8 and sorted(a8) == list(range(len(kinds)))


def test_mixed_engine_rejects_what_it_does_not_do():
    import ctypes as C
    from radiosonde_auto_rx_amd.engine import MixedEngine, SondeError, lib
    with pytest.raises(SondeError):
        MixedEngine([0.1, 0.2], ["rs41", "dfm"], 2_400_000, bits=32)
    eng = MixedEngine([0.1, 0.2], ["rs41", "dfm"], 2_400_000, max_chunk=240_000)
    assert eng.group_info(0)[0] == 41 and eng.group_info(1)[0] == 9
    assert eng.group_info(0)[1]["ring_len"] == eng.group_info(1)[1]["ring_len"]        # one y ring, one length
    assert lib().sonde_engine_restart_channel(eng._h, 0) < 0
    assert lib().sonde_engine_set_threshold(eng._h, C.c_float(0.5)) < 0
    eng.close()


def test_mixed_engine_with_m20_and_a_single_group():
    """M20 next to M10 and RS41 (sr 480 kHz), and an engine whose channels all carry the same type: one group, the shared launches with a single set of arguments"""
    from tools import synth
    from radiosonde_auto_rx_amd.engine import Engine, MixedEngine
    sr = 480_000
    kinds = ["m20", "rs41", "m10", "m20"]
    rng = np.random.default_rng(5)
    fqs = [synth.snap_fq(float(rng.uniform(-0.3, 0.3)), sr) for _ in kinds]
    caps = []
    for k, kd in enumerate(kinds):
        if kd == "m20":
            caps.append(synth.m10_capture(sr=sr, seconds=3.3, fq=fqs[k], noise_sigma=0.03, seed=40 + k, baud=9600.0, t_first=0.3 + 0.1 * k,
                                          frame_fn=lambda i, k=k: synth.m20_frame(i, fw=8, rng=np.random.default_rng(800 + 10 * k + i), good_checksum=(i != 1))))
        else:
            caps.append(_capture(kd, k, sr, 3.3, fqs[k]))
    x = np.stack(caps)

    def lines(eng, m20):
        out = {}
        for s0 in range(0, x.shape[1] // 2, sr):
            m = min(sr, x.shape[1] // 2 - s0) // 10 * 10
            eng.process_host(x[:, 2 * s0:2 * (s0 + m)] if isinstance(eng, MixedEngine) else x[sel][:, 2 * s0:2 * (s0 + m)])
        fin = True
        fr = eng.fetch_mxx(finish=fin, m20=True) if isinstance(eng, MixedEngine) else eng.fetch_mxx(finish=fin)
        for f in fr:
            out.setdefault(f["channel"], []).append((f["line"].rstrip(), f["mv_pos"], f.get("blk_ok")))
        return out
    eng = MixedEngine(fqs, kinds, sr, max_chunk=sr, max_frames=64)
    got = lines(eng, True)
    assert eng.group_info(0)[0] == 20 and eng.group_info(2)[0] == 10
    eng.close()
    sel = [0, 3]
    e1 = Engine([fqs[c] for c in sel], sr, sonde="m20", ecc=0, max_chunk=sr, max_frames=64)
    want = lines(e1, True)
    e1.close()
    assert sorted(got) == [0, 3] and {sel[c]: v for c, v in want.items()} == got and all(len(v) >= 2 for v in got.values())
    # one group only
    sel = [1]
    a = _per_channel(_run_mixed([fqs[1]] * 3, ["rs41"] * 3, np.ascontiguousarray(x[[1, 1, 1]]), sr, sr))
    b = _per_channel(_run_single([fqs[1]] * 3, ["rs41"] * 3, np.ascontiguousarray(x[[1, 1, 1]]), sr, sr))
    assert a == b and sorted(a) == [0, 1, 2] and a[0] == a[1] == a[2]


def test_mixed_engine_groups_of_one_type_with_different_options_taps_overflow_and_channel_end(bank24):
    """The C ABI directly: two RS41 groups (--ecc and --ecc2) beside a DFM group; read_tap and finish_channel go to the channel's group; a frame queue too small says so."""
    import ctypes as C
    from radiosonde_auto_rx_amd.engine import (lib, SondeCfg, SondeGroup, SondeFrame, SondeInfo, ABI_VERSION, SONDE_MIXED, LP_IQ, Engine, TAP_DECIM, TAP_BUFS, _chk)
    sr, fqs, x = bank24
    idx = [0, 4, 1, 6]                                             # rs41, rs41, dfm, rs41 of the bank
    kinds = [KINDS[i] for i in idx]
    assert kinds == ["rs41", "rs41", "dfm", "rs41"]
    fq = np.asarray([fqs[i] for i in idx], np.float64)
    xs = np.ascontiguousarray(x[idx][:, :4 * sr])
    L = lib()
    cfg = SondeCfg(ABI_VERSION, 0, 4, sr, 16, SONDE_MIXED, LP_IQ, 0, 0, 0, 0, 0.0, sr, 64, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0)
    groups = (SondeGroup * 3)(SondeGroup(41, 1, 0, 0, 0, 0, 0.0, 0), SondeGroup(9, 1, 0, 0, 0, 0, 0.0, 0), SondeGroup(41, 2, 0, 0, 0, 0, 0.0, 0))
    gof = np.asarray([0, 2, 1, 2], np.int32)                        # channel 0: rs41 --ecc; channels 1, 3: rs41 --ecc2; channel 2: dfm
    h = C.c_void_p()
    _chk(L.sonde_engine_create_mixed(C.byref(cfg), fq.ctypes.data_as(C.POINTER(C.c_double)), groups, 3, gof.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(h)))
    buf = (SondeFrame * 64)()
    got = {}
    for k in range(2):
        _chk(L.sonde_engine_process_host(h, np.ascontiguousarray(xs[:, 2 * k * sr:2 * (k + 1) * sr]).ctypes.data_as(C.c_void_p), sr, sr))
        n = _chk(L.sonde_engine_fetch_frames(h, buf, 64))
        for i in range(n):
            got.setdefault(buf[i].channel, []).append((buf[i].ecc, buf[i].mv_pos, bytes(buf[i].frame)))
    # the same channels on single-type engines with the group's ecc level
    for c, ecc in ((0, 1), (1, 2), (3, 2)):
        e = Engine([fq[c]], sr, ecc=ecc, max_chunk=sr, max_frames=16)
        fr = []
        for k in range(2):
            e.process_host(xs[c:c + 1, 2 * k * sr:2 * (k + 1) * sr])
            fr += e.fetch_frames()
        tap_want = e.read_tap(0, TAP_BUFS, 90000, 4000)
        e.close()
        assert len(fr) >= 1 and got.get(c) == [(f["ecc"], f["mv_pos"], f["frame"]) for f in fr], c
        out = np.zeros(4000, np.float32)
        _chk(L.sonde_engine_read_tap(h, c, TAP_BUFS, 90000, 4000, out.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(out, tap_want)                          # (read_tap reaches the channel's group: the same sliced stream to the bit)
    # end of one channel's stream: the DFM hit in progress on channel 2 is cut where the samples end
    _chk(L.sonde_engine_finish_channel(h, 2))
    from radiosonde_auto_rx_amd.engine import SondeDfmFrame
    dbuf = (SondeDfmFrame * 64)()
    nd = _chk(L.sonde_engine_fetch_dfm(h, dbuf, 64, 0))
    assert nd >= 1 and all(dbuf[i].channel == 2 for i in range(nd))
    assert L.sonde_engine_overflowed(h) == 0
    L.sonde_engine_destroy(h)
    # bad arguments
    h2 = C.c_void_p()
    bad = np.asarray([0, 3, 1, 2], np.int32)
    assert L.sonde_engine_create_mixed(C.byref(cfg), fq.ctypes.data_as(C.POINTER(C.c_double)), groups, 3, bad.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(h2)) < 0
    cfg.keep_soft = 1
    assert L.sonde_engine_create_mixed(C.byref(cfg), fq.ctypes.data_as(C.POINTER(C.c_double)), groups, 3, gof.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(h2)) < 0
