"""GPU parity of -i / --auto (polarity) in the IQ and FM-audio forms of rs41mod / dfm09mod: a header whose score has the
wrong sign is skipped, or flips the channel's polarity with --auto; sliced bits and soft values are negated while the
polarity is inverted (rs41mod.c:2887-2891,2933-2937; dfm09mod.c:1642-1645,1702-1705).  Captures with a mirrored spectrum
(Q negated, fq -> -fq) or negated FM audio.  Golden = stdout of the compiled reference (tools/make_golden.py INV_CASES);
text lines are compared exactly, including the cases where nothing may be decoded."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402
from golden_cases import need_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("name", sorted(make_golden.INV_CASES))
def test_cli_polarity_matches_reference(name):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    case = make_golden.INV_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    _, stdin, binary, args, _ = make_golden.inv_capture(case)
    r = subprocess.run([os.path.join(ROOT, "host", "bin", binary)] + args, input=stdin, capture_output=True, timeout=180)
    assert r.returncode == int(g["rc"]), r.stderr
    assert r.stderr.decode() == str(g["stderr"])
    assert [l.rstrip() for l in r.stdout.decode().splitlines()] == [l.rstrip() for l in g["lines"]]


def test_engine_polarity_is_per_channel():
    """Two channels of one engine, one mirrored: with --auto each channel settles on its own polarity and both decode.  Checked
    against the compiled reference per channel (`rs41mod -r --ecc2 --crc --auto --IQ fq --lpIQ` on the upright capture, the same with -fq on
    the mirrored one: identical stdout) and against the reference's soft bits of the mirrored capture sliced with inverted polarity."""
    from oracle import bind
    from radiosonde_auto_rx_amd.engine import Engine
    from tools import synth
    sr = 2_400_000
    fq = synth.snap_fq(0.1, sr)
    x = synth.rs41_capture(sr=sr, seconds=2.2, fq=fq, n_frames=2, t_first=0.1, noise_sigma=0.02, seed=91)
    xm = x.copy(); xm[1::2] = np.clip(-xm[1::2].astype(np.int32), -32768, 32767).astype(np.int16)
    eng = Engine([fq, -fq], sr, auto=True, keep_soft=True, max_chunk=sr, max_frames=8)
    fr = []
    n = len(x) // 2
    for s0 in range(0, n, sr // 2):
        s1 = min(n, s0 + sr // 2)
        eng.process_host(np.stack([x[2 * s0:2 * s1], xm[2 * s0:2 * s1]]))
        fr += eng.fetch_frames(with_soft=True)
    fr += eng.fetch_frames(with_soft=True, finish=True)
    eng.close()
    by = {c: [f for f in fr if f["channel"] == c] for c in (0, 1)}
    assert len(by[0]) == 2 and len(by[1]) == 2
    assert [f["line"] for f in by[0]] == [f["line"] for f in by[1]]
    assert all(f["mv"] > 0 for f in by[0]) and all(f["mv"] < 0 for f in by[1])
    if need_ref():
        for c, (cap, f0) in enumerate(((x, fq), (xm, -fq))):
            out, _, rc = bind.ref_run("rs41mod", ["-r", "--ecc2", "--crc", "--auto", "--IQ", repr(f0), "--lpIQ", "-", str(sr), "16"], cap)
            assert rc == 0 and [l.rstrip() for l in out.splitlines()] == [f["line"].rstrip() for f in by[c]], c
    # the same mirrored capture with -i on a one-channel engine: same positions and bit-identical soft values
    eng = Engine([-fq], sr, inv=True, keep_soft=True, max_chunk=sr, max_frames=8)
    fi = []
    for s0 in range(0, n, sr // 2):
        s1 = min(n, s0 + sr // 2)
        eng.process_host(xm[2 * s0:2 * s1])
        fi += eng.fetch_frames(with_soft=True)
    fi += eng.fetch_frames(with_soft=True, finish=True)
    eng.close()
    assert len(fi) == 2
    for a, b in zip(by[1], fi):
        assert a["mv_pos"] == b["mv_pos"] and np.array_equal(a["soft"], b["soft"])
    if need_ref():
        out, _, rc = bind.ref_run("rs41mod", ["-r", "--ecc2", "--crc", "-i", "--IQ", repr(-fq), "--lpIQ", "-", str(sr), "16"], xm)
        assert rc == 0 and [l.rstrip() for l in out.splitlines()] == [f["line"].rstrip() for f in fi]
