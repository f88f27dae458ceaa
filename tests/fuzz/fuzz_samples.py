"""Differential fuzzing of the SAMPLE forms of the nine decoder front ends against the compiled reference decoders on an MI355X: random decoder, random
input form (--IQ fq at several rates, --iq0/2/3, FM audio in a WAV; 16 / 8 / 32 bit), random filter / AFC / shift / threshold options, random noise,
frequency offset, polarity.  Every case = the same bytes and arguments into host/bin/<dec> and oracle/_ref/<dec>; stdout and exit code must agree.
    python tests/fuzz/fuzz_samples.py <seed> <seconds of wall clock> [keep_dir]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def degenerate_only(out_a: bytes, out_b: bytes) -> bool:
    """Digital silence inside a frame, rails clipped to constants, a constant on both rails: the discriminator / the tone sums are EXACTLY zero there, the
    reference's bits are the signs of what its recursive float sums have left over (demod_mod.c:778-808 keeps F1sum / F2sum recursively over the whole stream),
    the product's windowed sums give exact zeros.  Tolerated: the same number of lines, and every line that differs is a frame no block code
    accepts on either side (no [OK]; DFM: the blocks marked [OK] on either side are equal), or differs only in the count of repaired symbols by one."""
    import re
    la, lb = out_a.split(b"\n"), out_b.split(b"\n")
    if len(la) != len(lb):
        return False
    for u, v in zip(la, lb):
        if u == v:
            continue
        if b"[OK]" not in u and b"[OK]" not in v:
            continue
        bu, bv = re.split(rb"(\[OK\]|\[NO\]|\[KO\])", u), re.split(rb"(\[OK\]|\[NO\]|\[KO\])", v)
        # (a DFM block sliced from exact zeros is all ones — ties go to 1 — and all ones are a valid Hamming codeword: the product marks such a block [OK])
        if len(bu) == len(bv) and len(bu) >= 5 and all(x == y or (set(x.strip()) == {ord("F")} and n != b"[OK]")
                                                       for x, y, m, n in zip(bu[0::2], bv[0::2], bu[1::2] + [b""], bv[1::2] + [b""]) if m == b"[OK]" or n == b"[OK]"):
            continue
        if re.sub(rb"\(\d+\)", b"()", u) == re.sub(rb"\(\d+\)", b"()", v):
            cu, cv = [int(x) for x in re.findall(rb"\((\d+)\)", u)], [int(x) for x in re.findall(rb"\((\d+)\)", v)]
            if len(cu) == len(cv) and all(abs(x - y) <= 1 for x, y in zip(cu, cv)):
                continue
        return False
    return True


def one(rng, it, keep_dir):
    dec = list(DECS)[it % len(DECS)]
    D = DECS[dec]
    form = ["IQ", "IQ", "IQ", "iq0", "iq2", "iq3", "audio"][rng.integers(7)]
    # (rates that are no multiple of 48 kHz make the reference raise its IF rate to the next divisor: 250 k -> 50 k, 1.024 M -> 51.2 k, 1 M -> 50 k, 2.048 M -> 51.2 k)
    sr = int([48_000, 96_000, 240_000, 480_000, 250_000, 1_024_000, 1_000_000, 2_048_000, 300_000, 192_000][rng.integers(10)]) if form == "IQ" else 48_000
    if dec == "dfm09mod" and sr == 48_000 and form != "IQ":
        sr = 48_000
    fq = synth.snap_fq(float(rng.uniform(-0.3, 0.3)), sr) if (form == "IQ" and sr > 48_000) else 0.0
    ns = float([0.01, 0.03, 0.08, 0.15][rng.integers(4)])
    sd = int(rng.integers(1, 1 << 20))
    x = D["cap"](sr, fq, ns, sd)
    if rng.integers(3) == 0:                                        # a residual offset (the AFC of --dc unlocks beyond 1 kHz, demod_mod.c:1586-1599)
        z = (x[0::2] + 1j * x[1::2]) * np.exp(2j * np.pi * float(rng.uniform(-1500, 1500) if rng.integers(2) else rng.uniform(-600, 600)) / sr * np.arange(len(x) // 2))
        x = np.empty_like(x); x[0::2] = np.clip(np.round(z.real), -32768, 32767); x[1::2] = np.clip(np.round(z.imag), -32768, 32767)
    if rng.integers(5) == 0:
        x = x.copy(); x[1::2] = -x[1::2]                           # spectrum / polarity inversion
    anom = k = int(rng.integers(8))
    if k == 0:                                                     # digital silence in front (the reference's mp = -1 path, tests/test_gpu_silence.py)
        x = np.concatenate([np.zeros(2 * int(sr * float(rng.uniform(0.05, 0.4))), np.int16), x])
    elif k == 1:                                                   # ... in the middle
        p0 = 2 * int(rng.integers(len(x) // 8, len(x) // 2)); x = x.copy(); x[p0:p0 + 2 * int(sr * float(rng.uniform(0.01, 0.5)))] = 0
    elif k == 2:                                                   # a constant offset on both rails
        x = np.clip(x.astype(np.int32) + int(rng.integers(-6000, 6000)), -32768, 32767).astype(np.int16)
    elif k == 3:                                                   # clipping
        x = np.clip(x.astype(np.int32) * 3, -32768, 32767).astype(np.int16)
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
    if not ok and anom < 4 and ra.returncode == rb.returncode and degenerate_only(ra.stdout, rb.stdout):
        return True, dec, -1                                      # (counted apart: raw bits of frames no block code accepts, on degenerate input)
    if not ok:
        print("MISMATCH", dec, " ".join(args), "seed", sd, "noise", ns, "anomaly", {0: "silence in front", 1: "silence inside", 2: "offset", 3: "clipping"}.get(anom, "none"), "rc", ra.returncode, rb.returncode, flush=True)
        la, lb = ra.stdout.splitlines(), rb.stdout.splitlines()
        for u, v in zip(la, lb):
            if u != v:
                print(" OUR:", u[:160]); print(" REF:", v[:160])
                break
        else:
            print(" line counts", len(la), len(lb), ra.stderr[-160:])
        if keep_dir:
            os.makedirs(keep_dir, exist_ok=True)
            open(os.path.join(keep_dir, f"fail_{dec}_{it}.ours"), "wb").write(ra.stdout)
            open(os.path.join(keep_dir, f"fail_{dec}_{it}.ref"), "wb").write(rb.stdout)
        if keep_dir and len(data) < (20 << 20):
            open(os.path.join(keep_dir, f"fail_{dec}_{it}.bin"), "wb").write(data)
            open(os.path.join(keep_dir, f"fail_{dec}_{it}.args"), "w").write(" ".join(args))
    return ok, dec, len(ra.stdout)


def run(seed, budget_s, keep_dir=None):
    rng = np.random.default_rng(seed)
    t0, n, bad, silent, degen = time.time(), 0, 0, 0, 0
    while time.time() - t0 < budget_s:
        ok, dec, nout = one(rng, n, keep_dir)
        n += 1; bad += (not ok); silent += (nout == 0); degen += (nout == -1)
    print(f"cases {n}, mismatches {bad}, cases without output {silent}, cases in which only frames that no block code accepts differ (degenerate input: zeros / clipped / constant rails) {degen}")
    return bad


if __name__ == "__main__":
    sys.exit(min(255, run(int(sys.argv[1]), float(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else None)))
