"""debug: native rs41mod --ecc3 vs the seam (reference code on the same engine soft bits) vs the all-CPU reference"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
ECEF = dict(ecef_cm=(418833319, 85974133, 473346430))
tail = ["--IQ", "0.0", "--lpIQ", "-", "48000", "16"]
for ns, seed in ((0.47, 57), (0.5, 58)):
    x = synth.rs41_capture(sr=48_000, seconds=12.3, fq=0.0, noise_sigma=ns, frame_kw=ECEF, n_frames=12, t_first=0.15, seed=seed).tobytes()
    outs = {}
    for k, b in (("nat", "host/bin/rs41mod"), ("seam", "oracle/_ref/rs41mod_seam"), ("ref", "oracle/_ref/rs41mod")):
        outs[k] = subprocess.run([os.path.join(ROOT, b), "-r", "--ecc3", "--crc"] + tail, input=x, capture_output=True).stdout.decode().splitlines()
    print(ns, {k: len(v) for k, v in outs.items()})
    for i in range(len(outs["ref"])):
        row = [outs[k][i] if i < len(outs[k]) else "" for k in ("nat", "seam", "ref")]
        tags = [r[-12:] for r in row]
        d_ns = sum(a != b for a, b in zip(row[0], row[1])); d_sr = sum(a != b for a, b in zip(row[1], row[2])); d_nr = sum(a != b for a, b in zip(row[0], row[2]))
        print(i, tags, "nat-seam", d_ns, "seam-ref", d_sr, "nat-ref", d_nr)
