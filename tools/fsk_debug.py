"""GPU-box debug: batched 2-FSK modem vs the reference harness (oracle/_ref/libref_fsk.so)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tools import synth
from radiosonde_auto_rx_amd.fsk import FskModem, CS16
from oracle import bind

np.set_printoptions(precision=5, linewidth=220, suppress=True)


def run(name, iq, Fs, Rs, chunk, **kw):
    print("====", name)
    r = bind.ref_fsk_run(iq, Fs, Rs, P=kw.get("P", 8), nsym=kw.get("nsym", 50), mask=1 if kw.get("mask") else 0, tone_spacing=kw.get("mask") or 100,
                         lower=kw.get("lower"), upper=kw.get("upper"))
    md = FskModem(Fs, Rs, max_chunk=chunk, **kw)
    print(md.info, r["consts"], "ref frames", r["n"])
    n = len(iq) // 2
    sds, recs = [], []
    for s0 in range(0, n, chunk):
        md.process_host(iq[2 * s0:2 * min(n, s0 + chunk)])
        sd, rc = md.fetch(0)
        sds.append(sd); recs += rc
    sd = np.concatenate(sds)
    print("frames", len(recs), "kernel ms", md.kernel_ms())
    m = min(len(recs), r["n"])
    nin = np.array([x["nin"] for x in recs[:m]]); fe = np.array([x["f_est"] for x in recs[:m]])
    print("nin equal", np.array_equal(nin, r["nin"][:m]), "f_est equal", np.array_equal(fe, r["f_est"][:m]))
    for k in ("norm_rx_timing", "ppm", "EbNodB", "snr_est"):
        g = np.array([x[k] for x in recs[:m]])
        print(k, "max abs diff", float(np.abs(g - r[k][:m]).max()))
    d = sd[:m] - r["sd"][:m]
    print("sd rms", float(np.sqrt(np.mean(r["sd"][:m] ** 2))), "diff rms", float(np.sqrt(np.mean(d ** 2))), "max", float(np.abs(d).max()),
          "exact frac", float(np.mean(sd[:m] == r["sd"][:m])), "sign mismatches", int(np.sum((sd[:m] > 0) != (r["sd"][:m] > 0))))
    st = md.stats(0)
    print("Sf max diff", float(np.abs(st["Sf"] - r["Sf"]).max()), "Sf max", float(r["Sf"].max()), "samples", st["samples"])
    bad = np.where(nin != r["nin"][:m])[0]
    if len(bad):
        print("first nin mismatch at frame", bad[0], nin[max(0, bad[0] - 2):bad[0] + 3], r["nin"][max(0, bad[0] - 2):bad[0] + 3])


iq = synth.rs41_capture(sr=48000, seconds=3.0, fq=0.0, n_frames=2, t_first=0.4, noise_sigma=0.02, seed=7, f_offset_hz=900.0)
run("rs41 48k mask 5000 nsym 300 P 5", iq, 48000, 4800, 48000, P=5, nsym=300, mask=5000, lower=-20000, upper=20000)
run("rs41 48k peak est defaults", iq, 48000, 4800, 20000, P=10, nsym=50)
iq = synth.dfm_capture(sr=50000, seconds=2.0, fq=0.0, noise_sigma=0.02, seed=3)
run("dfm 50k", iq, 50000, 2500, 50000, P=10, nsym=50, lower=-15000, upper=15000)
