"""Differential fuzzing of host/bin/iq_dec (the front end alone: mixer, decimator, optional IF low-pass, FM discriminator, FM low-pass / --decFM, the writers)
against the compiled reference's iq_dec on an MI355X: random input rate (multiples of 48 kHz and rates that are not), input width 8 / 16 / 32 bit, --iq fq,
--IFbw, --lpIQ / --lpbw, --FM / --lpFM / --decFM, --dc, --noLUT, --min, output width --bo 8 / 16 / 32, --wav.
The two outputs must have the same length and header; float streams within 1e-5 RMS (the IF / FM stream tolerance of DESIGN.md §2 is 1e-6 RMS on streams of
0.3 rms) with no sample grossly off; integer streams within 0.5 steps RMS, no sample more than 32 steps off (the discriminator's angle where the IF amplitude
dips); a step of 2 * 0.8 where the angle sits at +-pi is the same angle.
    python tests/fuzz/fuzz_iqdec.py <seed> <seconds of wall clock>     -> prints every mismatch; exit code = number of mismatches (capped at 255)"""
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
    sr = int(rng.choice([48_000, 96_000, 240_000, 480_000, 960_000, 2_400_000, 250_000, 1_024_000, 1_000_000, 2_048_000]))
    fq = synth.snap_fq(float(rng.uniform(-0.4, 0.4)), sr) if sr > 48_000 else 0.0
    sec = float(rng.uniform(0.4, 1.6)) if sr > 1_000_000 else float(rng.uniform(0.8, 2.5))
    ns = float(rng.choice([0.01, 0.05, 0.2]))
    kind = str(rng.choice(["rs41", "dfm", "m10"]))
    sd = int(rng.integers(1, 1 << 30))
    if kind == "rs41":
        x = synth.rs41_capture(sr=sr, seconds=sec, fq=fq, seed=sd, noise_sigma=ns, t_first=0.05, f_offset_hz=float(rng.uniform(-2000, 2000)))
    elif kind == "dfm":
        x = synth.dfm_capture(sr=sr, seconds=sec, fq=fq, seed=sd, noise_sigma=ns, t_first=0.02)
    else:
        x = synth.m10_capture(sr=sr, seconds=sec, fq=fq, seed=sd, noise_sigma=ns, t_first=0.05)
    bits = int(rng.choice([16, 16, 8, 32]))
    data = (x if bits == 16 else synth.to_u8(x) if bits == 8 else synth.to_f32(x)).tobytes()
    a = []
    bo = int(rng.choice([0, 8, 16, 32]))
    if bo:
        a += ["--bo", str(bo)]
    if rng.integers(2) and sr > 48_000:
        a += ["--iq", repr(fq if rng.integers(4) else float(rng.uniform(-0.45, 0.45)))]
    if rng.integers(3) == 0:
        a += ["--IFbw", str(int(rng.choice([32, 48, 64, 96])))]
    k = int(rng.integers(4))
    if k == 0:
        a.append("--lpIQ")
    elif k == 1:
        a += ["--lpbw", "%.1f" % float(rng.uniform(4.0, 20.0))]
    k = int(rng.integers(5))
    if k == 0:
        a.append("--FM")
    elif k == 1:
        a.append("--lpFM")
    elif k == 2:
        a.append("--decFM")
    if rng.integers(4) == 0:
        a.append("--dc")
    if rng.integers(6) == 0:
        a.append("--noLUT")
    if rng.integers(5) == 0:
        a.append("--min")
    if rng.integers(5) == 0:
        a.append("--iqdc")
    wav = bool(rng.integers(3) == 0)
    if wav:
        a.append("--wav")
    args = a + ["-", str(sr), str(bits)]
    ra = subprocess.run(["host/bin/iq_dec"] + args, input=data, capture_output=True, timeout=300)
    rb = subprocess.run(["oracle/_ref/iq_dec"] + args, input=data, capture_output=True, timeout=300)
    why = None
    if ra.returncode != rb.returncode:
        why = f"rc {ra.returncode} / {rb.returncode}"
    elif len(ra.stdout) != len(rb.stdout):
        why = f"{len(ra.stdout)} / {len(rb.stdout)} bytes"
    elif ra.stdout != rb.stdout:
        hdr = (rb.stdout.find(b"data") + 8) if wav else 0          # (the header the reference writes, iq_dec.c:206-248)
        if ra.stdout[:hdr] != rb.stdout[:hdr]:
            why = "WAV header"
        else:
            w = bo if bo else 32                                   # (iq_dec.c:961: 32-bit float unless --bo says otherwise)
            fm = any(o in a for o in ("--FM", "--lpFM", "--decFM", "--dc"))
            dt = {8: np.uint8, 16: np.int16, 32: np.float32}[w]
            nb = (len(rb.stdout) - hdr) // (w // 8) * (w // 8)
            pa, pb = np.frombuffer(ra.stdout[hdr:hdr + nb], dt), np.frombuffer(rb.stdout[hdr:hdr + nb], dt)
            if w == 32:
                good = np.isfinite(pb)
                if not np.array_equal(np.isfinite(pa), good):
                    why = "non-finite samples in different places"
                else:
                    r = float(np.sqrt(np.mean(np.square(pb[good].astype(np.float64))))) if good.any() else 0.0
                    d = np.abs(pa[good].astype(np.float64) - pb[good])
                    wrap = fm & (np.abs(d - 1.6) < 1e-3)             # the discriminator's angle at +-pi comes out on either side: a step of 2 * 0.8
                    e_rms = float(np.sqrt(np.mean(np.square(d[~wrap])))) if (~wrap).any() else 0.0
                    gross = (d > 1e-2 * r + 1e-5) & ~wrap
                    lpf = any(o in a for o in ("--lpFM", "--decFM", "--dc"))   # behind the FM low-pass a step of 1.6 is spread over the taps around it: a few samples off by up to 0.4
                    if fm and lpf and gross.mean() < 5e-3 and float(d.max()) < 0.8:
                        gross[:] = False; e_rms = 0.0
                    # streams within 1e-6 RMS of the reference; single samples may be further off where the IF amplitude dips (the angle of a
                    # short vector), never grossly
                    if gross.any() or e_rms > 1e-5 * max(r, 0.3) / 0.3:
                        why = f"float samples: rms error {e_rms:.3g}, max {float(d.max()):.3g}, against rms {r:.3g}; {int(gross.sum())} of {len(d)} gross"
            else:
                d = np.abs(pa.astype(np.int32) - pb.astype(np.int32))
                wrap = fm & (d >= int(0.79 * (1 << (w - 1))))        # the same step in the integer forms
                e_rms = float(np.sqrt(np.mean(np.square(d[~wrap].astype(np.float64))))) if (~wrap).any() else 0.0
                lpf = any(o in a for o in ("--lpFM", "--decFM", "--dc"))
                if fm and lpf and (d > 32).mean() < 5e-3 and int(d.max()) < int(0.5 * (1 << (w - 1))):
                    d = np.minimum(d, 1); e_rms = 0.0
                if (d[~wrap] > 32).any() or e_rms > 0.5:
                    why = f"{w}-bit samples: rms error {e_rms:.3g} steps, max {int(d[~wrap].max()) if (~wrap).any() else 0}, {int(((d > 0) & ~wrap).sum())} of {len(d)} differ"
    if why:
        print("MISMATCH", " ".join(args), kind, "noise", ns, ":", why, flush=True)
        keep = os.path.join(ROOT, "gpurun_out", "fuzz_iqdec")
        os.makedirs(keep, exist_ok=True)
        if len(data) < (4 << 20):
            open(os.path.join(keep, f"fail_{it}.bin"), "wb").write(data)
        open(os.path.join(keep, f"fail_{it}.args"), "w").write(" ".join(args))
    return why is None, len(rb.stdout)


def main():
    seed, seconds = int(sys.argv[1]), float(sys.argv[2])
    rng = np.random.default_rng(seed)
    t0, n, bad, empty = time.time(), 0, 0, 0
    while time.time() - t0 < seconds:
        ok, nout = one(rng, n)
        n += 1; bad += 0 if ok else 1; empty += 1 if nout == 0 else 0
    print(f"fuzz_iqdec seed {seed}: cases {n}, mismatches {bad}, cases without output {empty}, {time.time() - t0:.0f} s")
    sys.exit(min(bad, 255))


if __name__ == "__main__":
    main()
