"""Differential fuzzing of host/bin/fsk_demod (the modem behind the C ABI) against the compiled reference's fsk_demod on an MI355X: random 2- / 4-FSK
geometry (symbol rate, samples per symbol 5 .. 40, oversampling P, frame length), tone spacing, estimator limits / mask estimator, input format (cs16, cu8,
real s16), amplitude from clipping down to a few LSB, noise, a stretch of digital silence in front or in the middle, hard or soft output, -i.
Hard decisions must be the same bytes; soft decisions the same count and within 2e-6 of their RMS (tests/test_gpu_fsk.py's tolerance: the timing estimate's
atan2f), NaNs in the same places.
    python tests/fuzz/fuzz_fsk.py <seed> <seconds of wall clock>     -> prints every mismatch; exit code = number of mismatches (capped at 255)"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from tools import synth  # noqa: E402


def one(rng, it):
    M = int(rng.choice([2, 2, 2, 4]))
    Rs = int(rng.choice([1200, 2400, 2500, 4800, 9600, 9616]))
    Ts = int(rng.choice([5, 8, 10, 10, 16, 20, 40]))
    Fs = Rs * Ts
    divs = [p for p in (2, 4, 5, 8, 10, 16, 20, 40) if Ts % p == 0 and p <= Ts]
    P = int(rng.choice(divs))
    nsym = int(rng.choice([50, 50, 100, 300]))
    shift = Rs * int(rng.choice([1, 1, 2]))
    span = (M - 1) * shift
    f_low = float(rng.uniform(-0.4 * Fs, 0.4 * Fs - span)) if 0.8 * Fs > span else -span / 2.0
    fmt = int(rng.choice([2, 2, 3, 1]))                            # 2 cs16, 3 cu8, 1 real s16
    if fmt == 1:
        f_low = abs(f_low) % max(1.0, 0.4 * Fs - span) + 0.02 * Fs
    nbits = int(rng.integers(3, 12)) * nsym * (2 if M == 4 else 1)
    amp = float(rng.choice([0.4, 0.4, 0.9, 1.3, 0.05, 0.002]))
    ns = amp * float(rng.choice([0.01, 0.05, 0.2, 0.5]))
    bits = np.random.default_rng(int(rng.integers(1 << 30))).integers(0, 2, nbits)
    x = synth.mfsk_capture(bits, Fs, Rs, M, f_low=f_low, shift=float(shift), amp=amp, noise_sigma=ns, seed=int(rng.integers(1 << 30)))
    k = int(rng.integers(5))
    if k == 0:                                                     # digital silence in front
        x = np.concatenate([np.zeros(2 * int(rng.integers(1, 4 * nsym * Ts)), np.int16), x])
    elif k == 1:                                                   # ... in the middle
        p = 2 * int(rng.integers(len(x) // 4, len(x) // 2)); x = x.copy(); x[p:p + 2 * int(rng.integers(Ts, 3 * nsym * Ts))] = 0
    if fmt == 3:
        data = synth.to_u8(x).tobytes()
    elif fmt == 1:
        data = np.ascontiguousarray(x[0::2]).tobytes()
    else:
        data = x.tobytes()
    a = {2: ["--cs16"], 3: ["--cu8"], 1: []}[fmt]
    soft = bool(rng.integers(3))
    if soft:
        a.append("-s")
    if rng.integers(3) == 0 and (M - 1) * shift < 0.45 * Fs:       # (a mask wider than the band reads behind the spectrum in the reference; the engine refuses it)
        a += ["--mask", str(shift)]
    if rng.integers(2):
        lo = int(f_low - rng.uniform(0.02, 0.2) * Fs); hi = int(f_low + span + rng.uniform(0.02, 0.2) * Fs)
        lo = max(lo, -Fs // 2 + 1) if fmt != 1 else max(lo, 1); hi = min(hi, Fs // 2 - 1)
        if hi - lo > span + 2 * Rs // 5:
            a += ["-b", str(lo), "-u", str(hi)]
    if rng.integers(5) == 0:
        a.append("-i")
    a += ["--nsym=%d" % nsym, "-p", str(P), str(M), str(Fs), str(Rs), "-", "-"]
    ra = subprocess.run(["host/bin/fsk_demod"] + a, input=data, capture_output=True, timeout=300)
    rb = subprocess.run(["oracle/_ref/fsk_demod"] + a, input=data, capture_output=True, timeout=300)
    why = None
    if ra.returncode != rb.returncode:
        why = f"rc {ra.returncode} / {rb.returncode}"
    elif not soft:
        if ra.stdout != rb.stdout:
            na, nb = len(ra.stdout), len(rb.stdout)
            d = sum(u != v for u, v in zip(ra.stdout, rb.stdout))
            # (a hard decision is the sign / the arg-max of soft values that agree to 1e-6 of their rms: one that sits on the edge may fall either way — seen once: 1 of 500)
            if na != nb or d > max(1, nb // 500):
                why = f"hard decisions: {na} / {nb} bytes, {d} differ"
    else:
        fa, fb = np.frombuffer(ra.stdout[:len(ra.stdout) // 4 * 4], np.float32), np.frombuffer(rb.stdout[:len(rb.stdout) // 4 * 4], np.float32)
        if len(ra.stdout) != len(rb.stdout):
            why = f"soft decisions: {len(fa)} / {len(fb)}"
        elif len(fb):
            nan_a, nan_b = ~np.isfinite(fa), ~np.isfinite(fb)
            if not np.array_equal(nan_a, nan_b):
                why = f"non-finite soft decisions in different places ({int(nan_a.sum())} / {int(nan_b.sum())})"
            else:
                ok = ~nan_b
                r = float(np.sqrt(np.mean(np.square(fb[ok].astype(np.float64))))) if ok.any() else 0.0
                d = np.abs(fa[ok].astype(np.float64) - fb[ok]) if ok.any() else np.zeros(1)
                n_far = int(np.sum(d > 2e-6 * r))                  # beyond the tolerance of tests/test_gpu_fsk.py
                n_gross = int(np.sum(d > 1e-3 * r))                # not a rounding difference any more
                if n_gross or n_far > max(1, len(d) // 1000):
                    why = f"soft decisions: max error {float(d.max()):.3g} against rms {r:.3g}; {n_far} of {len(d)} beyond 2e-6 rms, {n_gross} beyond 1e-3 rms"
    if why:
        print("MISMATCH", " ".join(a), f"fmt {fmt} amp {amp} noise {ns:.4g} f_low {f_low:.1f} shift {shift} silence {k}:", why, flush=True)
        keep = os.path.join(ROOT, "gpurun_out", "fuzz_fsk")
        os.makedirs(keep, exist_ok=True)
        open(os.path.join(keep, f"fail_{it}.bin"), "wb").write(data)
        open(os.path.join(keep, f"fail_{it}.args"), "w").write(" ".join(a))
    return why is None, len(rb.stdout)


def main():
    seed, seconds = int(sys.argv[1]), float(sys.argv[2])
    rng = np.random.default_rng(seed)
    t0, n, bad, empty = time.time(), 0, 0, 0
    while time.time() - t0 < seconds:
        ok, nout = one(rng, n)
        n += 1; bad += 0 if ok else 1; empty += 1 if nout == 0 else 0
    print(f"fuzz_fsk seed {seed}: cases {n}, mismatches {bad}, cases without output {empty}, {time.time() - t0:.0f} s")
    sys.exit(min(bad, 255))


if __name__ == "__main__":
    main()
