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
