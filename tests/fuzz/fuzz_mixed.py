"""Differential fuzzing of the mixed-type engine (sonde_engine_create_mixed) against the compiled reference decoders on an MI355X: a random number of
channels with random types in random order, random sample rate (2.4 Msps = the hand-scheduled decimator, 480 / 960 kHz = the templated one), 16- or 8-bit
samples, random noise / bit errors / frame phases / tuning, calls of random lengths (any multiple of the decimation, IQ-DC segment edges inside calls),
fetches 0 / 1 / 2 calls behind.  Every channel's text lines must be the stdout of oracle/_ref/{rs41mod -r --ecc2, dfm09mod -r --ecc, m10mod -r -v} on the
same bytes, line for line.
    python tests/fuzz/fuzz_mixed.py <seed> <seconds of wall clock>         -> prints every mismatch; exit code = number of mismatching channels (capped at 255)"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from tools import synth  # noqa: E402
from tests.test_gpu_mixed import REF, _fetch_all  # noqa: E402


def capture(kind, rng, sr, seconds, fq):
    sd = int(rng.integers(1, 1 << 30))
    ns = float(rng.choice([0.01, 0.03, 0.08, 0.15]))
    if kind == "rs41":
        return synth.rs41_capture(sr=sr, seconds=seconds, fq=fq, seed=sd, noise_sigma=ns, bit_errors=int(rng.choice([0, 0, 5, 12, 20, 40])), t_first=float(rng.uniform(0.02, 0.9)))
    if kind == "dfm":
        return synth.dfm_capture(sr=sr, seconds=seconds, fq=fq, noise_sigma=ns, seed=sd, bit_errors_per_frame=int(rng.integers(0, 4)), t_first=float(rng.uniform(0.01, 0.4)))
    return synth.m10_capture(sr=sr, seconds=seconds, fq=fq, noise_sigma=ns, seed=sd, t_first=float(rng.uniform(0.05, 0.9)),
                             frame_fn=lambda i: synth.m10_frame(i, rng=np.random.default_rng(sd + i), good_checksum=(i + sd) % 4 != 3))


def one(rng, it):
    from radiosonde_auto_rx_amd.engine import MixedEngine
    from oracle import bind
    sr = int(rng.choice([2_400_000, 480_000, 480_000, 960_000]))
    C = int(rng.integers(1, 9 if sr == 2_400_000 else 13))
    kinds = [str(rng.choice(["rs41", "dfm", "m10"])) for _ in range(C)]
    seconds = float(rng.uniform(1.6, 3.4))
    fqs = [synth.snap_fq(float(rng.uniform(-0.42, 0.42)), sr) for _ in kinds]
    x = np.stack([capture(kd, rng, sr, seconds, fqs[c]) for c, kd in enumerate(kinds)])
    bits = 16 if rng.random() < 0.75 else 8
    xin = x if bits == 16 else ((x.astype(np.int32) >> 8) + 128).astype(np.uint8)
    lag = int(rng.choice([0, 0, 1, 2]))
    max_chunk = int(rng.choice([sr, sr // 2, 2 * sr]))
    eng = MixedEngine(fqs, kinds, sr, max_chunk=max_chunk, max_frames=32 * C, bits=bits)
    D = eng.info["decM"]
    n, pos, got, present, calls = x.shape[1] // 2, 0, [], set(kinds), []
    while pos < n:
        take = min(int(rng.choice([max_chunk, max_chunk, int(rng.integers(D, max_chunk + 1)), int(rng.integers(D, 40 * D))])), n - pos) // D * D
        if take <= 0:
            break
        eng.process_host(xin[:, 2 * pos:2 * (pos + take)])
        pos += take
        calls.append(take)
        got += _fetch_all(eng, present, False, lag=lag)
    got += _fetch_all(eng, present, True)
    over = eng.overflowed()
    eng.close()
    lines = {}
    for f in got:
        lines.setdefault(f["channel"], []).append(f["line"].rstrip())
    bad = 0
    for c, kd in enumerate(kinds):
        exe, args = REF[kd]
        r = subprocess.run([os.path.join(bind.REFDIR, exe)] + args + ["--IQ", repr(fqs[c]), "--lpIQ", "-", str(sr), str(bits)], input=xin[c, :2 * pos].tobytes(), capture_output=True, timeout=600)
        want = [ln.rstrip() for ln in r.stdout.decode().splitlines()]
        if lines.get(c, []) != want or over:
            bad += 1
            have = lines.get(c, [])
            k = next((i for i in range(min(len(have), len(want))) if have[i] != want[i]), min(len(have), len(want)))
            print(f"MISMATCH it {it} channel {c} {kd} sr {sr} bits {bits} lag {lag} fq {fqs[c]!r} calls {calls[:12]}{'...' if len(calls) > 12 else ''} overflow {over}: "
                  f"{len(have)} lines against {len(want)}, first difference at line {k}")
            if k < len(have):
                print("   ours:", have[k][:200])
            if k < len(want):
                print("   ref: ", want[k][:200])
    return bad, C, sum(len(v) for v in lines.values())


def main():
    seed, seconds = int(sys.argv[1]), float(sys.argv[2])
    rng = np.random.default_rng(seed)
    t0, it, bad, chans, nlines = time.time(), 0, 0, 0, 0
    while time.time() - t0 < seconds:
        b, c, ln = one(rng, it)
        bad += b; chans += c; nlines += ln; it += 1
    print(f"fuzz_mixed seed {seed}: {it} engines, {chans} channels, {nlines} lines compared, {bad} mismatching channels in {time.time() - t0:.0f} s")
    sys.exit(min(bad, 255))


if __name__ == "__main__":
    main()
