"""GPU parity AT THE GEOMETRY bench.py TIMES: one engine, many 2.4 Msps channels, input resident in HBM and handed over with
`sonde_engine_process_device` in 1 s calls — i.e. the hand-scheduled decimator `k_mix_decimate50` with its channel -> XCD block mapping
(`ch = (slot / wgs_per_ch) * 8 + xcd`, more than 8 channels, a channel count that is not a multiple of 8), several tiles per wave with the
one-value carry between them, full-second launches that are one IQ-DC segment, the header search / frame sync of many channels per launch.

Every channel has its own capture: carrier, start time, noise level, on-air bit errors (up to the RS decoder's limit and beyond),
amplitude, IQ-DC offset.  Per channel the frames must equal
  * the CPU oracle (`ora_rs41_decode`, oracle/): text line and header position exactly, header score within 5e-6, soft bits within
    1e-5 RMS (BASELINE.json north_star);
  * the stdout of the compiled reference `rs41mod -r --ecc2 --crc --IQ fq --lpIQ - 2400000 16` (oracle/_ref) line for line.
Reference: demod_mod.c:463-504,737-754,639-648 (front end), rs41mod.c:2530-2545 (raw line).

SONDE_MD_G (engine test aid) sets the tiles per wave: 512 channels x 1 s — the bench — runs with 16, a 27-channel batch would pick 1.
"""
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest
from golden_cases import rms, need_ref

pytestmark = pytest.mark.gpu
SR = 2_400_000
NCH = 27
SECONDS = 2.2


def _spec(c):
    rng = np.random.default_rng(7000 + c)
    return dict(fq=float(rng.uniform(-0.45, 0.45)), t_first=float(rng.uniform(0.03, 0.5)), noise_sigma=float(rng.choice([0.004, 0.01, 0.03, 0.06, 0.1])),
                bit_errors=int(rng.choice([0, 0, 3, 8, 14, 22, 40])), amp=float(rng.uniform(0.15, 0.6)),
                dc=complex(rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02)) if c % 3 == 0 else 0.0, seed=300 + c, first_frame_no=50 * c + 7)


def _make(c):
    from tools import synth
    s = _spec(c)
    fq = synth.snap_fq(s["fq"], SR)
    x = synth.rs41_capture(sr=SR, seconds=SECONDS, fq=fq, t_first=s["t_first"], noise_sigma=s["noise_sigma"], bit_errors=s["bit_errors"],
                           amp=s["amp"], dc=s["dc"], seed=s["seed"], first_frame_no=s["first_frame_no"], sonde_id="B%07d" % c)
    return fq, x


@pytest.fixture(scope="module")
def batch():
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        out = list(ex.map(_make, range(NCH)))
    return [f for f, _ in out], np.stack([x for _, x in out])


def _run_batch(fqs, x, tiles_per_wave, chunk=SR):
    import torch
    from radiosonde_auto_rx_amd.engine import Engine
    n = x.shape[1] // 2
    n -= n % 50
    old = os.environ.get("SONDE_MD_G")
    if tiles_per_wave:
        os.environ["SONDE_MD_G"] = str(tiles_per_wave)
    try:
        dev = torch.from_numpy(x).to("cuda:0")                       # [NCH, 2 * samples] int16, resident
        eng = Engine(fqs, SR, keep_soft=True, max_chunk=SR, max_frames=8 * len(fqs))
        frames = []
        pos = 0
        while pos < n:
            take = min(chunk, n - pos)
            eng.process_device(dev.data_ptr() + 4 * pos, x.shape[1] // 2, take)      # channel stride in samples, like bench.py
            frames += eng.fetch_frames(with_soft=True)
            pos += take
        frames += eng.fetch_frames(with_soft=True, finish=True)
        eng.close()
    finally:
        if old is None:
            os.environ.pop("SONDE_MD_G", None)
        else:
            os.environ["SONDE_MD_G"] = old
    return frames, n


@pytest.mark.parametrize("tiles_per_wave", [16, 0])
def test_batch_of_2400k_channels_matches_oracle_and_reference(oracle, batch, tiles_per_wave):
    fqs, x = batch
    frames, n = _run_batch(fqs, x, tiles_per_wave)
    by = {c: [f for f in frames if f["channel"] == c] for c in range(NCH)}
    total = 0
    for c in range(NCH):
        o = oracle.ora_rs41_decode(x[c, :2 * n], SR, fq=fqs[c])
        got = by[c]
        assert len(got) == o["n"], (c, len(got), o["n"])
        for i, f in enumerate(got):
            assert f["line"] == o["lines"][i], (c, i)
            assert f["mv_pos"] == o["mv_pos"][i], (c, i)
            assert abs(f["mv"] - o["mv"][i]) < 5e-6, (c, i)
            nb = (f["nbytes"] - 8) * 8
            assert rms(f["soft"][:nb] - o["soft"][i][:nb]) < 1e-5, (c, i)
        if need_ref():
            out, _, rc = oracle.ref_run("rs41mod", ["-r", "--ecc2", "--crc", "--IQ", repr(fqs[c]), "--lpIQ", "-", str(SR), "16"], x[c, :2 * n])
            assert rc == 0 and [l.rstrip() for l in out.splitlines()] == [f["line"].rstrip() for f in got], c
        total += len(got)
    assert total == 2 * NCH                     # two frames per channel (a late start leaves the second one cut short: still printed at EOF)
    assert sum("[OK]" in f["line"] for f in frames) >= NCH and sum("[NO]" in f["line"] for f in frames) >= 6      # correctable and uncorrectable both occur


def test_batch_chunking_does_not_change_frames(batch):
    """The same batch in 0.5 s calls (launches that end inside an IQ-DC segment, P tail / carry across calls of the asm path)"""
    fqs, x = batch
    a, _ = _run_batch(fqs, x, 16)
    b, _ = _run_batch(fqs, x, 16, chunk=SR // 2)
    key = lambda f: (f["channel"], f["mv_pos"])
    a.sort(key=key); b.sort(key=key)
    assert [(f["channel"], f["mv_pos"], f["line"]) for f in a] == [(f["channel"], f["mv_pos"], f["line"]) for f in b]
    for fa, fb in zip(a, b):
        assert rms(fa["soft"] - fb["soft"]) < 2e-6
