"""GPU parity of the batched soft-bit framer on the device (include/sonde_fsk.h sonde_softin_dev_*: find_softbinhead, the RS41 bit loop,
rs41_ecc — the consumer half of auto_rx's `fsk_demod ... | rs41mod --softin -i`, auto_rx/autorx/decode.py:901-909) against the reference's own
pipe: the lines of `oracle/_ref/fsk_demod -s ... | oracle/_ref/rs41mod --softin -i -r --ecc2` recorded in tests/golden (tools/make_golden.py),
and `oracle/_ref/rs41mod --softin` run here on damaged, inverted and re-chunked soft-bit streams.  Frames must be equal line for line
(bytes, ECC verdict marker)."""
import os
import subprocess

import numpy as np
import pytest

from golden_cases import load_fsk, fsk_capture, need_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "rs41mod")


def _modem(case, n_channels=1):
    from radiosonde_auto_rx_amd.fsk import FskModem
    return FskModem(case["cap"]["sr"], case["Rs"], n_channels=n_channels, P=case["P"], nsym=case["nsym"], fmt=case["fmt"],
                    lower=case["lower"], upper=case["upper"], mask=case["mask"], max_chunk=case["cap"]["sr"])


def _ref_lines(soft, args):
    r = subprocess.run([REF] + args, input=np.ascontiguousarray(soft, np.float32).tobytes(), capture_output=True, timeout=120)
    return r.stdout.decode().splitlines()


def test_modem_to_frames_on_the_device_equals_the_reference_pipe():
    """fsk_demod | rs41mod --softin -i -r --ecc2, both halves on the device, the soft decisions never on the host"""
    from radiosonde_auto_rx_amd.fsk import SoftinDev
    name = "fsk_rs41_48k_mask"
    g = load_fsk(name)
    x, case = fsk_capture(name)
    sr = case["cap"]["sr"]
    X = np.stack([x, x, x])
    md = _modem(case, n_channels=3)
    sf = SoftinDev(3, ecc=2, inv=True)
    lines = {0: [], 1: [], 2: []}
    for s0 in range(0, X.shape[1] // 2, sr):
        md.process_host(X[:, 2 * s0:2 * (s0 + sr)])
        sf.push_fsk(md)
        for f in sf.fetch():
            lines[f["channel"]].append(f["line"])
    assert len(g["rs41_lines"]) >= 1
    # (the stream ends inside the last frame the reference prints: rs41mod emits the frame in progress at EOF; the batch framer keeps it pending)
    n = len(lines[0])
    assert n >= 1 and lines[0] == g["rs41_lines"][:n] and lines[1] == lines[0] and lines[2] == lines[0]
    c = sf.counts()
    assert c["frames"] == 3 * n and c["dropped"] == 0
    md.close(); sf.close()


def _stream(rng, nframes, flips, invert=False, gap=2000):
    """soft bits of `nframes` RS41 frames (a short preamble, the header, 312 whitened bytes with valid RS parity, then noise — the decoder reads 510 bytes
    behind every header) with `flips[i]` bits of frame i negated"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    out = [rng.normal(0, 0.3, 200).astype(np.float32)]
    for i in range(nframes):
        bits = synth.rs41_onair_bits(synth.rs41_frame(100 + i, rng=np.random.default_rng(int(rng.integers(1 << 30)))), preamble_bytes=8)
        s = (2.0 * bits.astype(np.float32) - 1.0) * rng.uniform(0.6, 1.4, len(bits)).astype(np.float32)
        idx = 128 + rng.choice(len(bits) - 128, size=flips[i], replace=False)              # behind preamble and header
        s[idx] = -s[idx]
        out += [s, rng.normal(0, 0.3, gap).astype(np.float32)]
    v = np.concatenate(out)
    return -v if invert else v


@pytest.mark.parametrize("case", ["clean", "damaged", "inverted_auto", "inverted_plain", "softinv"])
def test_soft_streams_in_device_memory_equal_reference_rs41mod(case):
    need_ref()
    import torch
    from radiosonde_auto_rx_amd.fsk import SoftinDev
    rng = np.random.default_rng({"clean": 1, "damaged": 2, "inverted_auto": 3, "inverted_plain": 4, "softinv": 5}[case])
    C = 5
    flips = {"clean": [0, 0, 0], "damaged": [8, 40, 120]}.get(case, [3, 0, 10])
    streams = [_stream(rng, 3, [int(f * (1 + c % 2)) for f in flips], invert=(case != "clean" and case != "damaged")) for c in range(C)]
    n = min(len(s) for s in streams)
    S = np.stack([s[:n] for s in streams])
    # our polarity conventions = the reference's options: -i sets gpx.option.inv, --auto lets a header of the other sign flip it, --softinv negates the stream
    kw, args = {"clean": (dict(inv=False), []), "damaged": (dict(inv=False), []),
                "inverted_auto": (dict(inv=False, auto=True), ["--auto"]),
                "inverted_plain": (dict(inv=True), ["-i"]),
                "softinv": (dict(inv=False, softinv=True), [])}[case]
    sf = SoftinDev(C, ecc=2, **kw)
    d = torch.from_numpy(S).cuda()
    got = {c: [] for c in range(C)}
    pos = 0
    while pos < n:                                            # calls of uneven length: the ring, a frame in progress and the bits of an unfinished byte carry over
        k = int(rng.choice([301, 1000, 4800, 4097, 77]))
        k = min(k, n - pos)
        chunk = d[:, pos:pos + k].contiguous()
        sf.push_device(chunk.data_ptr(), k, k)
        for f in sf.fetch():
            got[f["channel"]].append(f["line"])
        pos += k
    for c in range(C):
        soft_flag = ["--softinv"] if case == "softinv" else ["--softin"]
        ref = _ref_lines(S[c], soft_flag + args + ["-r", "--ecc2"])
        # the reference prints the frame in progress at EOF as well; every complete frame must be there, in order
        assert len(got[c]) >= 2 and got[c] == ref[:len(got[c])], (case, c)
    cnt = sf.counts()
    assert cnt["frames"] == sum(len(v) for v in got.values())
    if case == "damaged":
        assert cnt["repaired"] > 0 and cnt["ecc_ok"] < cnt["frames"]          # 120+ flipped bits are beyond the code
    sf.close()


# ---------------------------------------------------------------- DFM09 and M10: the other two thirds of BASELINE configs[3]

def _ref(binary, soft, args):
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", binary)] + args, input=np.ascontiguousarray(soft, np.float32).tobytes(), capture_output=True, timeout=120)
    return [l.rstrip() for l in r.stdout.decode().splitlines()]


@pytest.mark.parametrize("inv,key", [(True, "dfm_lines"), (False, "dfm_lines_noinv")])
def test_dfm_modem_to_frames_on_the_device_equals_the_reference_pipe(inv, key):
    """fsk_demod | dfm09mod --softin [-i] -r --ecc with both halves on the device: the recorded lines of the reference's own pipe (tests/golden/fsk_dfm_50k.npz)"""
    from radiosonde_auto_rx_amd.fsk import SoftinDev
    name = "fsk_dfm_50k"
    g = load_fsk(name)
    x, case = fsk_capture(name)
    sr = case["cap"]["sr"]
    X = np.stack([x, x])
    md = _modem(case, n_channels=2)
    sf = SoftinDev(2, kind="dfm", ecc=1, inv=inv)
    lines = {0: [], 1: []}
    for s0 in range(0, X.shape[1] // 2, sr):
        md.process_host(X[:, 2 * s0:2 * (s0 + sr)])
        sf.push_fsk(md)
        for f in sf.fetch_dfm():
            lines[f["channel"]].append(f["line"].rstrip())
    want = [str(l).rstrip() for l in g[key]]
    n = len(lines[0])
    assert n >= 4 and lines[0] == want[:n] and lines[1] == lines[0] and len(want) - n <= 1          # (the reference drops a partial frame at EOF: at most the one in progress is missing)
    md.close(); sf.close()


def _dfm_soft_stream(rng, nhits, flips, invert=False):
    """soft SYMBOLS of DFM frames (Manchester halves, 2 per bit) as fsk_demod -s delivers them: random conf / dat blocks with valid Hamming codewords, `flips` symbols negated per frame"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    out = [rng.normal(0, 0.3, 150).astype(np.float32)]
    for h in range(nhits):
        nfr = 9                                                                    # a header hit takes eight frames; the ninth needs the next search
        fb = np.concatenate([synth.dfm_frame_bits([int(v) for v in rng.integers(0, 16, 7)], [int(v) for v in rng.integers(0, 16, 13)], [int(v) for v in rng.integers(0, 16, 13)]) for _ in range(nfr)])
        sym = np.empty(2 * len(fb), np.float32); sym[0::2] = 1.0 - 2.0 * fb; sym[1::2] = 2.0 * fb - 1.0          # bit b -> halves (not b, b)
        sym *= rng.uniform(0.6, 1.4, len(sym)).astype(np.float32)
        for f in range(nfr):
            idx = 560 * f + 40 + rng.choice(500, size=flips, replace=False)
            sym[idx] = -sym[idx] * 0.3
        out += [sym, rng.normal(0, 0.3, 333).astype(np.float32)]
    v = np.concatenate(out)
    return -v if invert else v


@pytest.mark.parametrize("ecc,flips,invert", [(1, 0, False), (1, 6, False), (2, 14, False), (2, 10, True), (0, 3, False)])
def test_dfm_soft_streams_in_device_memory_equal_reference_dfm09mod(ecc, flips, invert):
    need_ref()
    import torch
    from radiosonde_auto_rx_amd.fsk import SoftinDev
    rng = np.random.default_rng(700 + 10 * ecc + flips)
    C = 4
    streams = [_dfm_soft_stream(rng, 3, flips + c, invert) for c in range(C)]
    n = min(len(s) for s in streams)
    S = np.stack([s[:n] for s in streams])
    eargs = {0: [], 1: ["--ecc"], 2: ["--ecc2"]}[ecc]
    sf = SoftinDev(C, kind="dfm", ecc=ecc, inv=False, auto=invert)
    d = torch.from_numpy(S).cuda()
    got = {c: [] for c in range(C)}
    pos = 0
    while pos < n:
        k = min(int(rng.choice([2500, 1111, 63, 5000, 560])), n - pos)
        chunk = d[:, pos:pos + k].contiguous()
        sf.push_device(chunk.data_ptr(), k, k)
        for f in sf.fetch_dfm():
            got[f["channel"]].append(f["line"].rstrip())
        pos += k
    for c in range(C):
        ref = _ref("dfm09mod", S[c], ["--softin", "-r"] + eargs + (["--auto"] if invert else []))
        assert len(got[c]) >= 16 and got[c] == ref[:len(got[c])] and len(ref) - len(got[c]) <= 1, (ecc, flips, c, got[c][:3], ref[:3])
    cnt = sf.counts()
    assert cnt["frames"] == sum(len(v) for v in got.values())
    if ecc and flips:
        assert cnt["repaired"] > 0
    sf.close()


def test_m10_modem_to_frames_on_the_device_equals_the_reference_pipe():
    """auto_rx's M10 pipe (decode.py:1120 `fsk_demod --cs16 -b -10000 -u 10000 -s -p 5 2 48080 9616 - - | m10mod --softin ...`), both halves on the device, against
    both halves of the compiled reference on the same capture (tools/caller_cases.py "m10": tones a symbol rate apart, as the modem's estimator assumes)"""
    need_ref()
    import sys
    sys.path.insert(0, ROOT)
    from tools import caller_cases as cc
    from radiosonde_auto_rx_amd.fsk import FskModem, SoftinDev
    x = cc.capture("m10")
    sr = 48080
    p1 = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "fsk_demod"), "--cs16", "-b", "-10000", "-u", "10000", "-s", "-p", "5", "2", "48080", "9616", "-", "-"],
                        input=x.tobytes(), capture_output=True, timeout=300)
    want = _ref("m10mod", np.frombuffer(p1.stdout, np.float32), ["--softin", "-r", "-v"])
    assert len(want) >= 3
    md = FskModem(sr, 9616, n_channels=2, P=5, nsym=50, lower=-10000, upper=10000, max_chunk=sr)
    sf = SoftinDev(2, kind="m10", ecc=0, inv=False)
    X = np.stack([x, x])
    lines = {0: [], 1: []}
    for s0 in range(0, X.shape[1] // 2, sr):
        md.process_host(X[:, 2 * s0:2 * (s0 + sr)])
        sf.push_fsk(md)
        for f in sf.fetch_m10():
            lines[f["channel"]].append(f["line"].rstrip())
    n = len(lines[0])
    assert n >= 3 and lines[0] == want[:n] and lines[1] == lines[0] and len(want) - n <= 1
    assert sf.counts()["ecc_ok"] >= 2 * (n - 1)
    md.close(); sf.close()


@pytest.mark.parametrize("invert", [False, True])
def test_m10_soft_streams_in_device_memory_equal_reference_m10mod(invert):
    need_ref()
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    from radiosonde_auto_rx_amd.fsk import SoftinDev
    rng = np.random.default_rng(900 + int(invert))
    C = 3
    streams = []
    for c in range(C):
        parts = [rng.normal(0, 0.3, 100 + 7 * c).astype(np.float32)]
        for k in range(3):
            sym = synth.m10_symbols(rng=np.random.default_rng(50 * c + k), data=synth.m10_frame(k, rng=np.random.default_rng(60 * c + k)))
            s = (2.0 * sym.astype(np.float32) - 1.0) * rng.uniform(0.6, 1.4, len(sym)).astype(np.float32)
            if k == 1:
                idx = 80 + rng.choice(len(s) - 100, size=4 + c, replace=False); s[idx] = -s[idx]              # a damaged frame: checksum [NO]
            parts += [s, rng.normal(0, 0.3, 2600 + 11 * k).astype(np.float32)]                                  # the skipped rest of the second and a bit more
        streams.append(np.concatenate(parts))
    n = min(len(s) for s in streams)
    S = np.stack([s[:n] for s in streams])
    if invert:
        S = -S
    sf = SoftinDev(C, kind="m10", ecc=0, inv=False)
    d = torch.from_numpy(np.ascontiguousarray(S)).cuda()
    got = {c: [] for c in range(C)}
    pos = 0
    while pos < n:
        k = min(int(rng.choice([9616, 1000, 251, 4808])), n - pos)
        chunk = d[:, pos:pos + k].contiguous()
        sf.push_device(chunk.data_ptr(), k, k)
        for f in sf.fetch_m10():
            got[f["channel"]].append(f["line"].rstrip())
        pos += k
    for c in range(C):
        ref = _ref("m10mod", S[c], ["--softin", "-r", "-v"])
        assert len(got[c]) >= 2 and got[c] == ref[:len(got[c])] and len(ref) - len(got[c]) <= 1, (c, got[c][:2], ref[:2])
    sf.close()


def _pipe_case(kind):
    """(capture as cs16 int16 pairs, sample rate, modem factory, consumer factory, fetch method) of auto_rx's pipe for one family"""
    import sys
    sys.path.insert(0, ROOT)
    from radiosonde_auto_rx_amd.fsk import FskModem, SoftinDev
    if kind == "m10":
        from tools import caller_cases as cc
        x, sr = cc.capture("m10"), 48080
        return (x, sr, lambda n: FskModem(sr, 9616, n_channels=n, P=5, nsym=50, lower=-10000, upper=10000, max_chunk=sr),
                lambda n: SoftinDev(n, kind="m10", ecc=0, inv=False), "fetch_m10")
    name = {"rs41": "fsk_rs41_48k_mask", "dfm": "fsk_dfm_50k"}[kind]
    x, case = fsk_capture(name)
    assert case["fmt"] != 1
    return (x, case["cap"]["sr"], lambda n: _modem(case, n_channels=n),
            (lambda n: SoftinDev(n, ecc=2, inv=True)) if kind == "rs41" else (lambda n: SoftinDev(n, kind="dfm", ecc=1, inv=True)), "fetch" if kind == "rs41" else "fetch_dfm")


@pytest.mark.parametrize("abort", [False, True])
@pytest.mark.parametrize("kind", ["rs41", "dfm", "m10"])
def test_submit_and_collect_give_the_frames_of_the_synchronous_calls(monkeypatch, capfd, kind, abort):
    """The two-halves calls in the order a pipelined caller uses them — wait (k - 1), collect (k - 2), submit_fsk (k - 1), submit_device (k): the consumer of a second runs
    on its own stream beside the modem's next launch (the modem keeps two launches' soft decisions).  Same frames, same order per channel, same tallies as process + push;
    also when the modem has to repeat a channel (test hook SONDE_FSK_TEST_ABORT: channel 1 gives up in every launch) — the wait repeats it before the consumer reads."""
    import torch
    x, sr, mk_modem, mk_cons, fetch = _pipe_case(kind)
    X = torch.from_numpy(np.stack([x, x, x])).cuda()
    n = X.shape[1] // 2

    def run(two_halves):
        md, sf = mk_modem(3), mk_cons(3)
        lines = {0: [], 1: [], 2: []}
        for s0 in range(0, n, sr):
            m = min(sr, n - s0)
            ptr = X.data_ptr() + 2 * s0 * X.element_size()
            if two_halves:
                if s0 > 0:
                    md.wait()
                    sf.collect()
                    sf.submit_fsk(md)               # the consumer over the launch before ...
                md.submit_device(ptr, n, m)         # ... beside this one
            else:
                md.process_device(ptr, n, m)
                sf.push_fsk(md)
            for f in getattr(sf, fetch)():
                lines[f["channel"]].append(f["line"].rstrip())
        if two_halves:
            md.wait(); sf.collect(); sf.submit_fsk(md); sf.collect()
            for f in getattr(sf, fetch)():
                lines[f["channel"]].append(f["line"].rstrip())
        c = sf.counts()
        sd = [md.fetch(k)[0] for k in range(3)]
        md.close(); sf.close()
        return lines, c, sd

    plain = run(False)
    capfd.readouterr()
    if abort:
        monkeypatch.setenv("SONDE_FSK_TEST_ABORT", "1")
    got = run(True)
    err = capfd.readouterr().err
    assert ("repeating them frame by frame" in err) == abort
    assert len(plain[0][0]) >= 1 and plain[0][1] == plain[0][0] and plain[0][2] == plain[0][0]
    assert got[0] == plain[0] and got[1] == plain[1]
    for k in range(3):
        assert np.array_equal(got[2][k], plain[2][k])


def test_submit_without_wait_is_waited_for_by_the_next_call():
    """any call of an engine with a launch in flight waits for it first: fetch right after submit returns that launch's soft decisions"""
    import torch
    x, sr, mk_modem, _mk, _f = _pipe_case("rs41")
    X = torch.from_numpy(np.stack([x, x])).cuda()
    a, b = mk_modem(2), mk_modem(2)
    a.process_device(X.data_ptr(), X.shape[1] // 2, sr)
    b.submit_device(X.data_ptr(), X.shape[1] // 2, sr)
    for k in range(2):
        sa, ra = a.fetch(k); sb, rb = b.fetch(k)
        assert len(sa) > 0 and np.array_equal(sa, sb) and ra == rb
    b.submit_device(X.data_ptr() + 2 * sr * X.element_size(), X.shape[1] // 2, sr)       # (a second submit waits for the first)
    b.submit_device(X.data_ptr() + 4 * sr * X.element_size(), X.shape[1] // 2, sr)
    a.process_device(X.data_ptr() + 2 * sr * X.element_size(), X.shape[1] // 2, sr)
    a.process_device(X.data_ptr() + 4 * sr * X.element_size(), X.shape[1] // 2, sr)
    assert np.array_equal(a.fetch(1)[0], b.fetch(1)[0])
    a.close(); b.close()


def test_modem_submitted_twice_before_collect_waits_for_the_consumer():
    """A caller that leaves the order of include/sonde_fsk.h — the modem submitted twice while a consumer call is still in flight: the second of those launches writes the
    very buffer of soft decisions the consumer was given.  The launch waits for the consumer on the device (sonde_fsk_dev_reader_done): the frames are
    those of the synchronous calls over the same seconds (the consumer is given seconds 0 and 2, not 1, in both runs)."""
    import torch
    x, sr, mk_modem, mk_cons, fetch = _pipe_case("rs41")
    nch = 48                                                   # (enough channels that the consumer's kernels take a while)
    X = torch.from_numpy(np.stack([x] * nch)).cuda()
    n = X.shape[1] // 2
    assert n >= 3 * sr
    ptr = lambda k: X.data_ptr() + 2 * k * sr * X.element_size()

    def lines_of(sf):
        out = {}
        for f in getattr(sf, fetch)(16 * nch):
            out.setdefault(f["channel"], []).append(f["line"].rstrip())
        return out

    md, sf = mk_modem(nch), mk_cons(nch)
    md.process_device(ptr(0), n, sr); sf.push_fsk(md)
    md.process_device(ptr(1), n, sr)
    md.process_device(ptr(2), n, sr); sf.push_fsk(md)
    want = lines_of(sf)
    md.close(); sf.close()

    md, sf = mk_modem(nch), mk_cons(nch)
    md.submit_device(ptr(0), n, sr); md.wait()
    sf.submit_fsk(md)                                          # the consumer of second 0 on its own stream ...
    md.submit_device(ptr(1), n, sr); md.wait()                 # ... the modem's next launch (the other buffer) ...
    md.submit_device(ptr(2), n, sr); md.wait()                 # ... and the one after it, which overwrites what the consumer reads — before any collect
    sf.collect()
    sf.submit_fsk(md); sf.collect()
    got = lines_of(sf)
    md.close(); sf.close()
    assert sorted(want) == list(range(nch)) and all(len(v) >= 1 for v in want.values())
    assert got == want


@pytest.mark.parametrize("kind", ["rs41", "dfm", "m10"])
def test_decoder_submitted_behind_the_modems_next_second_gives_the_same_frames(kind):
    """The order that keeps the modem's stream busiest — wait (k - 1), submit_device (k), collect (k - 2), submit_fsk_behind (k - 1): the decoder reads the launch BEFORE
    the one in flight (the modem's other buffer of soft decisions, the host's frame counts from before that launch was submitted), nothing waits for the launch in flight.
    Same frames, same order per channel, same soft decisions as process + push."""
    import torch
    x, sr, mk_modem, mk_cons, fetch = _pipe_case(kind)
    X = torch.from_numpy(np.stack([x, x, x])).cuda()
    n = X.shape[1] // 2

    def run(behind):
        md, sf = mk_modem(3), mk_cons(3)
        lines = {0: [], 1: [], 2: []}
        for s0 in range(0, n, sr):
            m = min(sr, n - s0)
            ptr = X.data_ptr() + 2 * s0 * X.element_size()
            if behind:
                if s0 > 0:
                    md.wait()
                md.submit_device(ptr, n, m)         # the modem's next second first ...
                if s0 > 0:
                    sf.collect()
                    sf.submit_fsk_behind(md)        # ... then the decoder over the second before it
            else:
                md.process_device(ptr, n, m)
                sf.push_fsk(md)
            for f in getattr(sf, fetch)():
                lines[f["channel"]].append(f["line"].rstrip())
        if behind:
            md.wait(); sf.collect(); sf.submit_fsk_behind(md); sf.collect()      # (no launch in flight: the last one's soft decisions)
            for f in getattr(sf, fetch)():
                lines[f["channel"]].append(f["line"].rstrip())
        c = sf.counts()
        sd = [md.fetch(k)[0] for k in range(3)]
        md.close(); sf.close()
        return lines, c, sd

    plain, got = run(False), run(True)
    assert len(plain[0][0]) >= 1 and plain[0][1] == plain[0][0] and plain[0][2] == plain[0][0]
    assert got[0] == plain[0] and got[1] == plain[1]
    for k in range(3):
        assert np.array_equal(got[2][k], plain[2][k])
