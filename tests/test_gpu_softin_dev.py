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
