"""Differential fuzzing of the single-type ENGINE (sonde_engine_create through engine.py) under the calls a library user makes — calls of any length, several channels
with different options shared, --dc (the AFC loop restarts channels inside a call), --iqdc, --min, --lpbw, 8- / 16- / 32-bit samples, IF-rate input forms, FM audio,
pipelined or not — against the compiled reference decoder's stdout on every channel's bytes.  (The CLIs feed fixed calls; tests/fuzz/fuzz_mixed.py does this for the mixed engine.)
    python tests/fuzz/fuzz_chunks.py <seed> <seconds of wall clock>     -> prints every mismatch; exit code = number of mismatching channels (capped at 255)"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from tools import synth  # noqa: E402

REF = {"rs41": "rs41mod", "dfm": "dfm09mod", "m10": "m10mod"}


def one(rng, it):
    from radiosonde_auto_rx_amd.engine import Engine
    kind = str(rng.choice(["rs41", "dfm", "m10"]))
    form = str(rng.choice(["IQ", "IQ", "IQ", "iq0", "iq2", "iq3", "audio"]))
    sr = int(rng.choice([480_000, 480_000, 960_000, 240_000, 2_400_000])) if form == "IQ" else 48_000
    C = int(rng.integers(1, 5 if sr > 1_000_000 else 9))
    dc = bool(rng.integers(3) == 0)
    iqdc = bool(form != "IQ" and rng.integers(3) == 0)
    opt_min = bool(form == "IQ" and rng.integers(5) == 0)
    lpbw = round(float(rng.uniform(5.0, 20.0)), 1) if rng.integers(4) == 0 else 0.0
    lp_iq = bool(lpbw or rng.integers(3) > 0)
    lp_fm = bool(rng.integers(5) == 0)
    eng_lp_fm = lp_fm or (form == "IQ" and dc)                  # the decoders' own rule (rs41mod.c:2747, dfm09mod.c:1475, m10mod.c:1320): --IQ with --dc turns the FM low-pass on
    bits = int(rng.choice([16, 16, 8, 32]))
    if form == "audio":
        bits = int(rng.choice([16, 16, 8])); iqdc = False; opt_min = False; lpbw = 0.0; lp_iq = False
    ecc = {"rs41": int(rng.choice([1, 1, 2])), "dfm": int(rng.choice([0, 1, 2])), "m10": 0}[kind]      # (rs41mod never runs without its decoder: rs41mod.c:2756 raises 0 to 1)
    seconds = float(rng.uniform(1.8, 3.3))
    fqs = [synth.snap_fq(float(rng.uniform(-0.4, 0.4)), sr) if form == "IQ" else 0.0 for _ in range(C)]
    caps = []
    for c in range(C):
        sd = int(rng.integers(1, 1 << 30)); ns = float(rng.choice([0.01, 0.03, 0.08])); off = float(rng.uniform(-1500, 1500)) if dc else 0.0
        if kind == "rs41":
            caps.append(synth.rs41_capture(sr=sr, seconds=seconds, fq=fqs[c], seed=sd, noise_sigma=ns, bit_errors=int(rng.choice([0, 0, 8, 20])), t_first=float(rng.uniform(0.02, 0.9)), f_offset_hz=off))
        elif kind == "dfm":
            caps.append(synth.dfm_capture(sr=sr, seconds=seconds, fq=fqs[c] + off / sr, noise_sigma=ns, seed=sd, bit_errors_per_frame=int(rng.integers(0, 3)), t_first=float(rng.uniform(0.01, 0.4))))
        else:
            caps.append(synth.m10_capture(sr=sr, seconds=seconds, fq=fqs[c], noise_sigma=ns, seed=sd, t_first=float(rng.uniform(0.05, 0.9)), f_offset_hz=off,
                                          frame_fn=lambda i, sd=sd: synth.m10_frame(i, rng=np.random.default_rng(sd + i), good_checksum=(i + sd) % 4 != 3)))
    n = min(len(c) for c in caps) // 2
    x = np.stack([c[:2 * n] for c in caps])
    if form == "audio":                                           # FM audio (a WAV for the decoder, the PCM for the engine), one audio channel
        pcm = np.stack([synth.fm_audio(c, gain=float(rng.uniform(0.15, 0.45))) for c in x])
        x = pcm; n = pcm.shape[1]
        xin = pcm if bits == 16 else synth.to_u8(pcm.reshape(-1)).reshape(pcm.shape)
    else:
        xin = x if bits == 16 else synth.to_u8(x.reshape(-1)).reshape(x.shape) if bits == 8 else synth.to_f32(x.reshape(-1)).reshape(x.shape)
    per = 1 if form == "audio" else 2
    max_chunk = int(rng.choice([sr, sr // 2, 2 * sr]))
    pipeline = bool(rng.integers(2)) and not dc and form != "audio"      # (the pipelined form is the base-rate / IF-rate engines'; an FM-audio engine refuses it)
    try:
      eng = Engine(fqs, sr, sonde=kind, ecc=ecc, lp_iq=lp_iq, lp_fm=eng_lp_fm, lpiq_bw=int(round(lpbw * 1e3)), opt_dc=dc, opt_min=opt_min, bits=bits, iqdc=iqdc,
                 iq_mode={"IQ": 5, "iq0": 1, "iq2": 2, "iq3": 3, "audio": 5}[form], audio=(form == "audio"), max_chunk=max_chunk, max_frames=64 * C, pipeline=pipeline)
    except Exception as ex:
        print(f"REFUSED it {it}: {kind} {form} sr {sr} bits {bits} dc {dc} iqdc {iqdc} min {opt_min} lpbw {lpbw} lp_iq {lp_iq} lp_fm {lp_fm} pipeline {pipeline}: {ex}", flush=True)
        return 0, 0, 0
    D = eng.info["decM"]
    fetch = {"rs41": lambda fin: eng.fetch_frames(finish=fin), "dfm": lambda fin: eng.fetch_dfm(finish=fin), "m10": lambda fin: eng.fetch_mxx(finish=fin)}[kind]
    pos, lines, calls = 0, {}, []
    while pos < n:
        take = min(int(rng.choice([max_chunk, max_chunk, int(rng.integers(D, max_chunk + 1)), int(rng.integers(D, 60 * D))])), n - pos) // D * D
        if take <= 0:
            break
        eng.process_host(xin[:, per * pos:per * (pos + take)])
        pos += take; calls.append(take)
        for f in fetch(False):
            lines.setdefault(f["channel"], []).append(f["line"].rstrip())
    for f in fetch(True):
        lines.setdefault(f["channel"], []).append(f["line"].rstrip())
    over = eng.overflowed()
    eng.close()
    args = ["-r"] + ({"rs41": [[], [] if rng.integers(2) else ["--ecc"], ["--ecc2"]], "dfm": [[], ["--ecc"], ["--ecc2"]], "m10": [["-v"]]}[kind][ecc if kind != "m10" else 0])
    bad = 0
    for c in range(C):
        a = list(args)
        if form == "audio":
            if lp_fm:
                a.append("--lpFM")
            if dc:
                a.append("--dc")
            data = synth.wav_bytes(np.ascontiguousarray(xin[c, :pos]), sr, 1, bits)
            r = subprocess.run([os.path.join("oracle", "_ref", REF[kind])] + a, input=data, capture_output=True, timeout=600)
            want = [ln.rstrip() for ln in r.stdout.decode().splitlines()]
            have = lines.get(c, [])
            if have != want or over:
                bad += 1
                k = next((i for i in range(min(len(have), len(want))) if have[i] != want[i]), min(len(have), len(want)))
                print(f"MISMATCH it {it} channel {c}/{C}: {REF[kind]} {' '.join(a)} <WAV {sr} Hz {bits} bit>  pipeline {pipeline} calls {calls[:10]} overflow {over}: {len(have)} lines against {len(want)}, first difference at line {k}", flush=True)
                if k < len(have):
                    print("   ours:", have[k][:160])
                if k < len(want):
                    print("   ref: ", want[k][:160])
            continue
        a += ["--IQ", repr(fqs[c])] if form == "IQ" else ["--" + form]
        if opt_min:
            a.append("--min")
        if lpbw:
            a += ["--lpbw", "%.1f" % lpbw]
        elif lp_iq:
            a.append("--lpIQ")
        if lp_fm:
            a.append("--lpFM")
        if dc:
            a.append("--dc")
        if iqdc:
            a.append("--iqdc")
        a += ["-", str(sr), str(bits)]
        r = subprocess.run([os.path.join("oracle", "_ref", REF[kind])] + a, input=np.ascontiguousarray(xin[c, :2 * pos]).tobytes(), capture_output=True, timeout=600)
        want = [ln.rstrip() for ln in r.stdout.decode().splitlines()]
        have = lines.get(c, [])
        if have != want and not over and len(have) == len(want):
            # a raw bit whose soft value sits at the float streams' noise floor may fall either way (tests/test_gpu_lowsnr.py): the same text behind the hex, <= 2 bits
            def close(u, v):
                hu, hv = u.split(" ", 1) + [""], v.split(" ", 1) + [""]
                try:
                    return u == v or (hu[1] == hv[1] and len(hu[0]) == len(hv[0]) and bin(int(hu[0], 16) ^ int(hv[0], 16)).count("1") <= 2 and "[OK]" not in u)
                except ValueError:
                    return False
            if all(close(u, v) for u, v in zip(have, want)):
                print(f"(noise-floor bit) it {it} channel {c}/{C}: {REF[kind]} {' '.join(a)}", flush=True)
                continue
        if have != want or over:
            bad += 1
            k = next((i for i in range(min(len(have), len(want))) if have[i] != want[i]), min(len(have), len(want)))
            print(f"MISMATCH it {it} channel {c}/{C}: {REF[kind]} {' '.join(a)}  pipeline {pipeline} calls {calls[:10]}{'...' if len(calls) > 10 else ''} overflow {over}: "
                  f"{len(have)} lines against {len(want)}, first difference at line {k}", flush=True)
            if k < len(have):
                print("   ours:", have[k][:160])
            if k < len(want):
                print("   ref: ", want[k][:160])
    return bad, C, sum(len(v) for v in lines.values())


def main():
    seed, seconds = int(sys.argv[1]), float(sys.argv[2])
    rng = np.random.default_rng(seed)
    t0, it, bad, chans, nlines = time.time(), 0, 0, 0, 0
    while time.time() - t0 < seconds:
        b, c, ln = one(rng, it)
        bad += b; chans += c; nlines += ln; it += 1
    print(f"fuzz_chunks seed {seed}: {it} engines, {chans} channels, {nlines} lines compared, {bad} mismatching channels in {time.time() - t0:.0f} s")
    sys.exit(min(bad, 255))


if __name__ == "__main__":
    main()
