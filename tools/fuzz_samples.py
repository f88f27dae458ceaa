"""Differential fuzzing of the SAMPLE forms of the nine decoder front ends against the compiled reference decoders on an MI355X: random decoder, random
input form (--IQ fq at several rates, --iq0/2/3, FM audio in a WAV; 16 / 8 / 32 bit), random filter / AFC / shift / threshold options, random noise,
frequency offset, polarity.  Every case = the same bytes and arguments into host/bin/<dec> and oracle/_ref/<dec>; stdout and exit code must agree.
    python tools/fuzz_samples.py <seed> <seconds of wall clock> [keep_dir]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from tools import synth  # noqa: E402

env = dict(os.environ, SONDE_JSN_VERSION="oracle")
DECS = {
    "rs41mod": dict(cap=lambda sr, fq, ns, sd: synth.rs41_capture(sr=sr, seconds=3.3, fq=fq, noise_sigma=ns, seed=sd, n_frames=3, t_first=0.12, bit_errors=int(sd % 4)),
                    opts=[["-r", "--ecc2", "--crc"], ["-r", "--ecc"], ["--ecc2", "--json", "--ptu2"], ["-r", "--ecc3"], ["-v", "--ptu", "--ecc2"], ["-r", "--ecc4"]], lpbw=(4.6, 24.0)),
    "dfm09mod": dict(cap=lambda sr, fq, ns, sd: synth.dfm_capture(sr=sr, seconds=2.6, fq=fq, noise_sigma=ns, seed=sd, bit_errors_per_frame=int(sd % 3)),
                     opts=[["-r", "--ecc"], ["-r", "--ecc2"], ["-r"], ["-vv", "--ecc", "--json", "--dist", "--auto"]], lpbw=(4.6, 24.0)),
    "m10mod": dict(cap=lambda sr, fq, ns, sd: synth.m10_capture(sr=sr, seconds=3.2, fq=fq, noise_sigma=ns, seed=sd, frame_fn=lambda j: synth.m10_frame(j, rng=np.random.default_rng(sd + j))),
                   opts=[["-r"], ["-r", "-v"], ["--json", "--ptu", "-vvv"], ["-v", "--ptu"]], lpbw=(4.6, 48.0)),
    "m20mod": dict(cap=lambda sr, fq, ns, sd: synth.m10_capture(sr=sr, seconds=3.2, fq=fq, noise_sigma=ns, seed=sd, baud=9600.0, type_bytes=(0x45, 0x20), frame_fn=lambda j: synth.m20_frame(j)),
                   opts=[["-r"], ["-r", "-v"], ["--json", "--ptu", "-vvv"]], lpbw=(4.6, 48.0)),
    "lms6Xmod": dict(cap=lambda sr, fq, ns, sd: synth.lms6_capture(sr=sr, seconds=3.6, fq=fq, noise_sigma=ns, seed=sd),
                     opts=[["-r", "--ecc"], ["--vit", "--ecc", "--json"], ["--lms6", "--vit2", "--ecc3"], ["-r"]], lpbw=(4.6, 24.0)),
    "meisei100mod": dict(cap=lambda sr, fq, ns, sd: synth.meisei_capture(sr=sr, seconds=4.2, fq=fq, noise_sigma=ns, seed=sd, variant="rs11g" if sd % 2 else "ims100"),
                         opts=[["--ecc"], ["-r", "--ecc", "-v"], ["--json", "--ptu", "--ecc"], ["-r"]], lpbw=(4.6, 32.0)),
    "imet54mod": dict(cap=lambda sr, fq, ns, sd: synth.imet54_capture(sr=sr, seconds=3.6, fq=fq, noise_sigma=ns, seed=sd, check=["std", "cont", "none"][sd % 3]),
                      opts=[["--ecc"], ["-r", "--ecc"], ["--ecc", "--json", "--ptu"], ["--ecc", "-v", "--ptu"]], lpbw=(4.6, 24.0)),
    "mp3h1mod": dict(cap=lambda sr, fq, ns, sd: synth.mrz_capture(sr=sr, seconds=4.4, fq=fq, noise_sigma=ns, seed=sd, latlon=False),
                     opts=[[], ["-r"], ["-vv", "--ptu"], ["--json", "--ptu"]], lpbw=(4.6, 32.0)),
    "mts01mod": dict(cap=lambda sr, fq, ns, sd: synth.mts01_capture(sr=sr, seconds=3.6, fq=fq, noise_sigma=ns, seed=sd),
                     opts=[[], ["-r"], ["-v", "--json"]], lpbw=(4.6, 48.0)),
}


def one(rng, it, keep_dir):
    dec = list(DECS)[it % len(DECS)]
    D = DECS[dec]
    form = ["IQ", "IQ", "IQ", "iq0", "iq2", "iq3", "audio"][rng.integers(7)]
    sr = int([48_000, 96_000, 240_000, 480_000][rng.integers(4)]) if form == "IQ" else 48_000
    if dec == "dfm09mod" and sr == 48_000 and form != "IQ":
        sr = 48_000
    fq = synth.snap_fq(float(rng.uniform(-0.3, 0.3)), sr) if (form == "IQ" and sr > 48_000) else 0.0
    ns = float([0.01, 0.03, 0.08, 0.15][rng.integers(4)])
    sd = int(rng.integers(1, 1 << 20))
    x = D["cap"](sr, fq, ns, sd)
    if form != "IQ" and rng.integers(3) == 0:                       # a residual offset the IF-rate forms have to live with
        z = (x[0::2] + 1j * x[1::2]) * np.exp(2j * np.pi * float(rng.uniform(-600, 600)) / sr * np.arange(len(x) // 2))
        x = np.empty_like(x); x[0::2] = np.clip(np.round(z.real), -32768, 32767); x[1::2] = np.clip(np.round(z.imag), -32768, 32767)
    if rng.integers(5) == 0:
        x = x.copy(); x[1::2] = -x[1::2]                           # spectrum / polarity inversion
    a = list(D["opts"][rng.integers(len(D["opts"]))])
    if form == "audio":
        if "--ecc3" in a or "--ecc4" in a or "--vit2" in a:
            a = [o for o in a if o not in ("--ecc3", "--ecc4", "--vit2")]
        pcm = synth.fm_audio(x, gain=float(rng.uniform(0.1, 0.5)))
        if rng.integers(4) == 0:
            data = synth.wav_bytes(synth.to_u8(pcm), sr, 1, 8)
        else:
            data = synth.wav_bytes(pcm, sr, 1, 16)
        tail = []
        if rng.integers(3) == 0:
            a.append("--dc")
        if rng.integers(4) == 0:
            a.append("--lpFM")
    else:
        bits = int([16, 16, 8, 32][rng.integers(4)])
        data = (x if bits == 16 else synth.to_u8(x) if bits == 8 else synth.to_f32(x)).tobytes()
        if form == "IQ":
            a += ["--IQ", repr(fq)]
            if rng.integers(5) == 0 and sr > 48_000:
                a.append("--min")
            if rng.integers(6) == 0 and "--dc" not in a:
                a.append("--noLUT")
        else:
            a.append("--" + form)
            if rng.integers(3) == 0:
                a.append("--iqdc")
        k = rng.integers(4)
        if k == 0:
            a.append("--lpIQ")
        elif k == 1:
            a += ["--lpbw", "%.1f" % float(rng.uniform(D["lpbw"][0] + 0.2, min(D["lpbw"][1], 20.0)))]
        if rng.integers(4) == 0:
            a.append("--lpFM")
        if rng.integers(3) == 0 and "--noLUT" not in a:
            a.append("--dc")
        tail = ["-", str(sr), str(bits)]
    if rng.integers(5) == 0:
        a += ["-d", str(int(rng.integers(-2, 3)))]
    if rng.integers(6) == 0 and dec != "mp3h1mod":
        a += ["--ths", "%.2f" % float(rng.uniform(0.55, 0.85))]
    args = a + tail
    ra = subprocess.run(["host/bin/" + dec] + args, input=data, capture_output=True, env=env, timeout=120)
    rb = subprocess.run(["oracle/_ref/" + dec] + args, input=data, capture_output=True, timeout=120)
    ok = ra.returncode == rb.returncode and ra.stdout == rb.stdout
    if not ok:
        print("MISMATCH", dec, " ".join(args), "seed", sd, "noise", ns, "rc", ra.returncode, rb.returncode, flush=True)
        la, lb = ra.stdout.splitlines(), rb.stdout.splitlines()
        for u, v in zip(la, lb):
            if u != v:
                print(" OUR:", u[:160]); print(" REF:", v[:160])
                break
        else:
            print(" line counts", len(la), len(lb), ra.stderr[-160:])
        if keep_dir:
            open(os.path.join(keep_dir, f"fail_{dec}_{it}.bin"), "wb").write(data)
            open(os.path.join(keep_dir, f"fail_{dec}_{it}.args"), "w").write(" ".join(args))
    return ok, dec, len(ra.stdout)


def run(seed, budget_s, keep_dir=None):
    rng = np.random.default_rng(seed)
    t0, n, bad, silent = time.time(), 0, 0, 0
    while time.time() - t0 < budget_s:
        ok, dec, nout = one(rng, n, keep_dir)
        n += 1; bad += (not ok); silent += (nout == 0)
    print(f"cases {n}, mismatches {bad}, cases without output {silent}")
    return bad


if __name__ == "__main__":
    sys.exit(min(255, run(int(sys.argv[1]), float(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else None)))
