"""debug: native fsk_demod vs compiled reference on the 4-FSK mask case, frame by frame"""
import os, sys, subprocess, json, re
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth
rng = np.random.default_rng(4)
bits = rng.integers(0, 2, 2 * 50 * 200)
x = synth.mfsk_capture(bits, 48000, 2400, 4, f_low=-3600.0, shift=2400.0, noise_sigma=0.12, seed=9).tobytes()
args = ["--cs16", "-p", "5", "--stats=100", "--mask", "2400", "-s", "4", "48000", "2400", "-", "-"]
out = {}
for k, b in (("nat", "host/bin/fsk_demod"), ("ref", "oracle/_ref/fsk_demod")):
    r = subprocess.run([os.path.join(ROOT, b)] + args, input=x, capture_output=True)
    sd = np.frombuffer(r.stdout, np.float32).reshape(-1, 100)
    st = [json.loads(re.sub(r"-?nan", "NaN", l)) for l in r.stderr.decode().splitlines() if l.startswith("{")]
    out[k] = (sd, st)
a, b = out["nat"], out["ref"]
print(a[0].shape, b[0].shape, len(a[1]), len(b[1]))
d = np.abs(a[0] - b[0]).max(axis=1)
for f in range(len(d)):
    if d[f] > 1e-2 or f < 4:
        sa, sb = a[1][f], b[1][f]
        print(f, d[f], [sa[k] for k in ("samples", "f1_est", "f2_est", "f3_est", "f4_est", "ppm", "EbNodB")], [sb[k] for k in ("samples", "f1_est", "f2_est", "f3_est", "f4_est", "ppm", "EbNodB")])
        bad = np.where(np.abs(a[0][f] - b[0][f]) > 1e-2)[0]
        print("   bad idx", bad[:20], a[0][f][bad[:6]], b[0][f][bad[:6]])
