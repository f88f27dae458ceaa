#!/usr/bin/env python3
"""First-light debug on the GPU box: print engine-vs-oracle differences stage by stage."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import bind
from radiosonde_auto_rx_amd import engine as E
from golden_cases import capture, load, rms

name = sys.argv[1] if len(sys.argv) > 1 else "rs41_480k_clean"
x, fq, sr = capture(name)
g = load(name)
eng = E.Engine([fq], sr, keep_soft=True, max_chunk=sr)
print("info", eng.info)
D = eng.info["decM"]
n = min(len(x) // 2 // D * D, sr)
t = time.time(); eng.process_host(x[:2 * n]); eng.sync(); print("process", time.time() - t)
nif = n // D
s = bind.ora_streams(x[:2 * n], sr, fq=fq)
print("oracle consts", s["consts"])
for tap, key in ((E.TAP_IFIQ, "iq"), (E.TAP_FM, "fm"), (E.TAP_BUFS, "bufs")):
    a = eng.read_tap(0, tap, 0, nif)
    d = a - s[key]
    print(key, "rms", rms(d), "max", float(np.abs(d).max()), "at", int(np.argmax(np.abs(d).reshape(len(d), -1).max(axis=1))),
          "sig", rms(s[key]), "first", a[:3].ravel(), s[key][:3].ravel())
frames = eng.fetch_frames(with_soft=True)
rest = len(x) // 2 - n
pos = n
while rest >= D:
    take = min(sr, rest) // D * D
    eng.process_host(x[2 * pos:2 * (pos + take)]); frames += eng.fetch_frames(with_soft=True)
    pos += take; rest -= take
o = bind.ora_rs41_decode(x, sr, fq=fq)
print("oracle frames", o["n"], o["mv"], o["mv_pos"], [l[-12:] for l in o["lines"]])
print("engine frames", len(frames), [(f["mv"], f["mv_pos"], f["line"][-12:]) for f in frames])
for i, f in enumerate(frames):
    if i < o["n"]:
        print(i, "line equal", f["line"] == o["lines"][i], "soft rms", rms(f["soft"] - o["soft"][i]), "floor", float(g["floor_soft"]))
for k in ("mix_decimate", "if_chain", "header_corr", "framesync"):
    print(k, eng.kernel_ms(k))
