"""Differential fuzzing of host/bin/dft_detect (the scanner behind the C ABI, matrix-core prefilter in front of the exact correlation) against the
compiled reference's dft_detect on an MI355X: random sonde type, input form (--iq at 48 / 96 kHz, --IQ fq at 0.48 / 2.4 Msps, FM audio in a WAV), random
amplitude (down to 2 % of full scale) and noise (scores on both sides of the thresholds), frequency offset, polarity, a quiet or loud stretch beside the
signal, random options (--bw, --dc, -t, -d2, -c, -v, -L, --min, --ths).  stdout and exit code must agree.
    python tests/fuzz/fuzz_scan.py <seed> <seconds of wall clock>     -> prints every mismatch; exit code = number of mismatches (capped at 255)"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from tools import synth  # noqa: E402


def capture(kind, rng, sr, fq, amp, ns, off):
    sd = int(rng.integers(1, 1 << 30))
    sec = float(rng.uniform(2.2, 4.2))
    if kind == "rs41":
        return synth.rs41_capture(sr=sr, seconds=sec, fq=fq, seed=sd, noise_sigma=ns, amp=amp, f_offset_hz=off, t_first=float(rng.uniform(0.05, 0.9)))
    if kind == "dfm":
        return synth.dfm_capture(sr=sr, seconds=sec, fq=fq + off / sr, noise_sigma=ns, seed=sd, amp=amp, t_first=float(rng.uniform(0.02, 0.5)))
    if kind == "m10":
        return synth.m10_capture(sr=sr, seconds=sec, fq=fq, noise_sigma=ns, seed=sd, amp=amp, f_offset_hz=off, t_first=float(rng.uniform(0.05, 0.9)))
    if kind == "m20":
        return synth.m10_capture(sr=sr, seconds=sec, fq=fq, noise_sigma=ns, seed=sd, amp=amp, f_offset_hz=off, baud=9600.0, type_bytes=(0x45, 0x20), frame_fn=lambda j: synth.m20_frame(j))
    if kind == "imet":
        x = synth.imet_capture(sr=sr, seconds=sec, f_offset_hz=off, amp=amp, noise_sigma=ns, seed=sd, space_hz=float(rng.choice([2200.0, 2400.0])))
        if fq:
            z = (x[0::2] + 1j * x[1::2]) * np.exp(2j * np.pi * fq * np.arange(len(x) // 2))
            x = np.empty_like(x); x[0::2] = np.clip(np.round(z.real), -32768, 32767); x[1::2] = np.clip(np.round(z.imag), -32768, 32767)
        return x
    if kind == "lms6":
        return synth.lms6_capture(sr=sr, seconds=sec, fq=fq, noise_sigma=ns, seed=sd, amp=amp)
    if kind == "noise":
        g = np.random.default_rng(sd)
        n = int(sr * sec)
        return np.clip(np.round(ns * 32767 * g.standard_normal(2 * n)), -32768, 32767).astype(np.int16)
    return synth.family_capture(kind, sr=sr, seconds=sec, fq=fq, amp=amp, noise_sigma=ns, seed=sd, f_offset_hz=off, t_first=float(rng.uniform(0.1, 0.9)), invert=bool(rng.integers(2)))


KINDS = ["rs41", "rs41", "dfm", "dfm", "m10", "m10", "m20", "imet", "lms6", "rs92mod", "imet54mod", "mp3h1mod", "mts01mod", "meisei100mod", "noise"]


def one(rng, it):
    kind = KINDS[int(rng.integers(len(KINDS)))]
    form = str(rng.choice(["iq", "iq", "IQ", "IQ", "audio"]))
    sr = int(rng.choice([48_000, 48_000, 96_000])) if form == "iq" else int(rng.choice([480_000, 480_000, 2_400_000])) if form == "IQ" else 48_000
    fq = synth.snap_fq(float(rng.uniform(-0.35, 0.35)), sr) if form == "IQ" else 0.0
    amp = float(rng.choice([0.5, 0.5, 0.2, 0.06, 0.02]))
    ns = amp * float(rng.choice([0.02, 0.06, 0.15, 0.3, 0.5, 0.8]))
    off = float(rng.uniform(-3000, 3000)) if rng.integers(2) else 0.0
    try:
        x = capture(kind, rng, sr, fq, amp, ns, off)
    except TypeError:                                              # (a generator without amp / f_offset_hz)
        return True, kind, 0
    if rng.integers(4) == 0:                                       # a stretch without carrier in front: full-scale discriminator noise beside the signal
        n0 = int(sr * float(rng.uniform(0.05, 0.4)))
        g = np.random.default_rng(it)
        x = np.concatenate([np.clip(np.round(float(rng.choice([0.0, 1e-3, 0.01])) * 32767 * g.standard_normal(2 * n0)), -32768, 32767).astype(np.int16), x])
    k = int(rng.integers(10))
    if k == 0:                                                     # digital silence somewhere inside (an all-zero window later in the stream: mp = -1 with a valid position)
        p0 = 2 * int(rng.integers(len(x) // 8, len(x) // 2)); x = x.copy(); x[p0:p0 + 2 * int(sr * float(rng.uniform(0.05, 0.6)))] = 0
    elif k == 1:                                                   # the stream ends in silence
        x = np.concatenate([x, np.zeros(2 * int(sr * float(rng.uniform(0.1, 0.5))), np.int16)])
    elif k == 2:                                                   # clipping
        x = np.clip(x.astype(np.int32) * 4, -32768, 32767).astype(np.int16)
    if rng.integers(5) == 0:
        x = x.copy(); x[1::2] = -x[1::2]
    a = []
    if rng.integers(2):
        a.append("-v")
    if rng.integers(4) == 0:
        a.append("-c")
    if rng.integers(5) == 0:
        a.append("-d2")
    if rng.integers(5) == 0:
        a.append("-L")
    if rng.integers(3) == 0:
        a += ["-t", str(int(rng.integers(1, 5)))]
    if rng.integers(6) == 0:
        a += ["--ths", "%.2f" % float(rng.uniform(0.4, 0.9))]
    if form == "audio":
        pcm = synth.fm_audio(x, gain=float(rng.uniform(0.1, 0.5)))
        data = synth.wav_bytes(pcm, sr, 1, 16)
        args = a
    else:
        bits = int(rng.choice([16, 16, 16, 8]))
        data = (x if bits == 16 else synth.to_u8(x)).tobytes()
        if form == "IQ":
            a += ["--IQ", repr(fq)]
            if rng.integers(5) == 0:
                a.append("--min")
        else:
            a.append("--iq")
        if rng.integers(2):
            a += ["--bw", str(int(rng.choice([8, 12, 15, 20, 32])))]
        if rng.integers(2):
            a.append("--dc")
        args = a + ["-", str(sr), str(bits)]
    ra = subprocess.run(["host/bin/dft_detect"] + args, input=data, capture_output=True, timeout=300)
    rb = subprocess.run(["oracle/_ref/dft_detect"] + args, input=data, capture_output=True, timeout=300)
    ok = ra.returncode == rb.returncode and ra.stdout == rb.stdout
    if not ok and k < 3:
        # digital silence inside / at the end, or rails clipped to constants: where the input is exactly constant the discriminator works on what float rounding leaves of
        # it (|z| ~ 1e-8 behind the decimator), and the IMET check's spectra / a template's score over such a stretch are noise that differs between any two
        # implementations (`degenerate input`): counted apart
        print("DEGENERATE", kind, " ".join(args), ["silence inside", "ends in silence", "clipping"][k], "rc", ra.returncode, rb.returncode, ra.stdout[:80], rb.stdout[:80], flush=True)
        return True, kind, -1
    if not ok:
        print("MISMATCH", kind, " ".join(args), "amp", amp, "noise", ns, "off", off, "rc", ra.returncode, rb.returncode, flush=True)
        la, lb = ra.stdout.decode(errors="replace").splitlines(), rb.stdout.decode(errors="replace").splitlines()
        for u, v in zip(la, lb):
            if u != v:
                print("  OUR:", u[:160]); print("  REF:", v[:160])
                break
        else:
            print("  line counts", len(la), len(lb), ra.stderr[-200:])
        keep = os.path.join(ROOT, "gpurun_out", "fuzz_scan")
        os.makedirs(keep, exist_ok=True)
        open(os.path.join(keep, f"fail_{it}.bin"), "wb").write(data)
        open(os.path.join(keep, f"fail_{it}.args"), "w").write(" ".join(args))
    return ok, kind, len(ra.stdout)


def main():
    seed, seconds = int(sys.argv[1]), float(sys.argv[2])
    rng = np.random.default_rng(seed)
    t0, n, bad, silent, degen = time.time(), 0, 0, 0, 0
    while time.time() - t0 < seconds:
        ok, kind, nout = one(rng, n)
        n += 1; bad += 0 if ok else 1; silent += 1 if nout == 0 else 0; degen += 1 if nout == -1 else 0
    print(f"fuzz_scan seed {seed}: cases {n}, mismatches {bad}, cases without a detection {silent}, differences on degenerate input (exactly constant stretches) {degen}, {time.time() - t0:.0f} s")
    sys.exit(min(bad, 255))


if __name__ == "__main__":
    main()
